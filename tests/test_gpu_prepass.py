"""-m gpu: the HIP viewer prepass (m2s_prepass, k_prepass) through the C ABI, against
  * the oracle (oracle/m2s_oracle_prepass.c, itself bit-identical to the reference's shader) on the same inputs, and
  * what the REFERENCE produced (tests/golden/ref_host/prepass_*: GaussiansPrepass.cpp + gaussianSplattingPrepassCS.glsl
    executed by oracle/_ref/ref_prepass_check).
Bar: survivor count and order exact; every float BIT-IDENTICAL (the kernel keeps the shader's operation order, IEEE
division and square root, no contraction), NaN matching NaN — except the colour of the two debug render modes that go
through library functions (mode 1: exp, mode 3: sin), compared within 1e-4 / on the unit circle as noted below."""
import os

import numpy as np
import pytest

import prepass_cases
import refhost
from mesh2splat_amd import synth
from mesh2splat_amd.converter import Converter

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_host")
CASES = prepass_cases.cases()


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same_bits(a, b):
    return (bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))


def assert_prepass_matches(got, want, mode, what):
    gk, gq, gd = got
    wk, wq, wd = want
    assert gk == wk, f"{what}: {gk} survivors, expected {wk}"
    ok = same_bits(gq, wq)
    if mode == 1:
        # colour.rgb = clamp(exp(-20 * d)): device expf vs libm; parity tolerance 1e-4 relative (+ tiny absolute floor)
        c = slice(8, 11)
        assert np.allclose(gq[:, c], wq[:, c], rtol=1e-4, atol=1e-30, equal_nan=True), f"{what}: depth colour"
        ok[:, c] = True
    if mode == 3:
        # colour.rgb = fract(sin(x) * 43758.5453): one ulp of sin moves the result by ~3e-3 and can wrap it around 1 -> 0.
        # Compare on the unit circle with the amplified tolerance.
        c = slice(8, 11)
        d = np.abs(gq[:, c] - wq[:, c])
        d = np.minimum(d, 1.0 - d)
        assert np.nanmax(d, initial=0.0) < 2e-2, f"{what}: hash colour differs by {np.nanmax(d)}"
        ok[:, c] = True
    assert ok.all(), f"{what}: quads differ at {np.argwhere(~ok)[:6].tolist()}"
    assert same_bits(gd, wd).all(), f"{what}: depths differ"


@pytest.fixture(scope="module")
def conv(hiplib):
    c = Converter(0)
    yield c
    c.close()


@pytest.mark.parametrize("name", [c[0] for c in CASES])
def test_hip_prepass_matches_reference_golden(conv, name):
    import torch
    p = dict(CASES)[name]
    with open(os.path.join(GOLD, "prepass_records.bin"), "rb") as f:
        rec = np.frombuffer(f.read(), np.float32).reshape(-1, 24).copy()
    with open(os.path.join(GOLD, f"prepass_{name}.out.bin"), "rb") as f:
        ref = refhost.parse_prepass_output(f.read())
    got = conv.prepass(p, records=torch.from_numpy(rec).cuda())
    assert_prepass_matches(got, ref, p.render_mode, f"golden {name}")


@pytest.mark.parametrize("name", [c[0] for c in CASES])
def test_hip_prepass_matches_oracle(conv, oracle, name):
    """More records than the golden set, including hostile ones, and a count that is not a multiple of 64."""
    import torch
    p = dict(CASES)[name]
    rec = np.concatenate([prepass_cases.base_records(oracle, 14, 64), prepass_cases.hostile_records(2048)])[:-13]
    got = conv.prepass(p, records=torch.from_numpy(rec).cuda())
    assert_prepass_matches(got, oracle.prepass(p, rec), p.render_mode, name)


def test_prepass_of_the_last_conversion_and_device_depth(conv, oracle):
    """The usual call sequence: convert, then prepass the context's own records; depth image handed over on the device."""
    import torch
    scene = synth.cube_sphere(40, tex_size=64)
    R = 256
    conv.upload_scene(scene)
    conv.set_max_gaussians(0)
    total = conv.convert(R)
    rec = conv.download()
    from dataclasses import replace
    p = replace(dict(CASES)["depth_test"], resolution_target=R)
    want = oracle.prepass(p, rec)
    got = conv.prepass(p)
    assert_prepass_matches(got, want, 0, "last conversion, host depth")
    assert 0 < got[0] < total
    pd = replace(p, mesh_depth=torch.from_numpy(np.ascontiguousarray(p.mesh_depth)).cuda())
    assert_prepass_matches(conv.prepass(pd), want, 0, "last conversion, device depth")
    # results stay addressable on the device
    assert conv.prepass(p, download=False) == want[0]
    assert conv.device_quads != 0 and conv.device_prepass_depths != 0


def canonical_rows(q, d):
    """Rows (quad + depth) as uint32 with every NaN set to one pattern, sorted lexicographically: order-free comparison."""
    rows = np.concatenate([q, d[:, None]], axis=1).astype(np.float32)
    u = rows.view(np.uint32).copy()
    u[np.isnan(rows)] = 0x7FC00000
    return u[np.lexsort(u.T[::-1])]


@pytest.mark.parametrize("name", ["colour", "depth_test", "model_trs", "inside"])
def test_prepass_arrival_order_yields_the_same_set(conv, oracle, name):
    """arrival_order = 1 is the reference's own contract (atomic append): same survivors, same values, any order; quads
    and depths stay paired."""
    import torch
    from dataclasses import replace
    p = replace(dict(CASES)[name], arrival_order=True)
    rec = np.concatenate([prepass_cases.base_records(oracle, 14, 64), prepass_cases.hostile_records(2048)])[:-13]
    wk, wq, wd = oracle.prepass(p, rec)
    d_rec = torch.from_numpy(rec).cuda()
    for _ in range(3):
        gk, gq, gd = conv.prepass(p, records=d_rec)
        assert gk == wk
        assert np.array_equal(canonical_rows(gq, gd), canonical_rows(wq, wd))


def test_sort_prepass_is_radix_sort_pass(conv, oracle):
    """RadixSortPass (RadixSortPass.cpp:8-90) on the prepass output: key = raw bits of the view-space depth, ascending,
    stable (LSD radix), six-vec4 gather.  Against numpy's stable argsort of the same keys on the oracle's prepass."""
    import torch
    p = dict(CASES)["colour"]
    rec = np.concatenate([prepass_cases.base_records(oracle, 20, 96), prepass_cases.hostile_records(4096)])
    rec = np.concatenate([rec, rec[:5000]])                       # duplicates: equal keys exercise stability
    wk, wq, wd = oracle.prepass(p, rec)
    gk, gq, gd = conv.prepass(p, records=torch.from_numpy(rec).cuda())
    assert gk == wk
    order = np.argsort(wd.view(np.uint32), kind="stable")
    conv.set_profiling(True)
    sq = conv.sort_prepass()
    conv.set_profiling(False)
    assert sq.shape == (wk, 24)
    assert same_bits(sq, wq[order]).all()
    assert conv.last_sort_prepass_ms > 0.0
    # negative view-space z (in front of the camera) has the sign bit set: raw-bit ascending order puts the positive
    # (behind the camera but inside the guard band) depths first, then the negative ones from nearest (-0.0...) to farthest
    keys = sq[:, 15].copy()                                       # conic.w = -z
    assert (np.diff((-keys).view(np.uint32).astype(np.int64)) >= 0).all()
    # a new prepass invalidates the sorted buffer; nothing to sort -> n = 0
    far = rec.copy()
    far[:, 0:3] = 1e6
    assert conv.prepass(p, records=torch.from_numpy(far).cuda())[0] == 0
    assert conv.sort_prepass(download=False) == 0


@pytest.mark.parametrize("fmt,has_pbr", [(0, False), (1, True)])
def test_load_ply_then_prepass_and_sort(conv, oracle, fmt, has_pbr):
    """The reference's LoadPly flow (guiRendererConcreteMediator.cpp:30-41): parsers::loadPlyFile -> gaussian buffer ->
    format 1 -> prepass -> radix sort.  Input: .ply files written by the REFERENCE (tests/golden/ref_host/ref_fmt*.ply)."""
    from dataclasses import replace
    from mesh2splat_amd import gltf_io
    rec, pbr = gltf_io.read_ply(os.path.join(GOLD, f"ref_fmt{fmt}.ply"))
    assert bool(pbr) == has_pbr
    conv.upload_records(rec)
    assert np.array_equal(conv.download().view(np.uint32), rec.view(np.uint32))
    view, proj = prepass_cases.default_camera((640, 360))
    p = replace(dict(CASES)["ply_classic"], ply_has_pbr=bool(pbr), render_mode=0)
    # the sample records come from a unit-size soup: look at it from nearby
    import camera
    p = replace(p, view_mat=camera.look_at((0.5, 0.4, 2.5), (0.5, 0.5, 0.0)))
    want = oracle.prepass(p, rec)
    assert want[0] > 0
    assert_prepass_matches(conv.prepass(p), want, 0, f"loaded ply fmt {fmt}")
    sq = conv.sort_prepass()
    assert same_bits(sq, want[1][np.argsort(want[2].view(np.uint32), kind="stable")]).all()
    if fmt == 0:                                   # an empty .ply: everything downstream sees zero records
        conv.upload_records(rec[:0])
        assert conv.prepass(p)[0] == 0 and conv.sort_prepass(download=False) == 0 and conv.download().shape[0] == 0


def test_prepass_edge_cases(conv, oracle):
    import torch
    p = dict(CASES)["colour"]
    one = prepass_cases.base_records(oracle, 6, 24)[:1]
    # nothing / one record / exactly one wave / one wave + 1 / everything culled / nothing culled
    for n in (1, 63, 64, 65, 257, 511, 512, 513, 1024, 1537):       # around wave (64) and workgroup (512) boundaries
        rec = np.tile(one, (n, 1))
        rec[:, 0:3] += np.linspace(0, 0.05, n, dtype=np.float32)[:, None]
        assert_prepass_matches(conv.prepass(p, records=torch.from_numpy(rec).cuda()), oracle.prepass(p, rec), 0, f"n={n}")
    rec = np.tile(one, (300, 1))
    rec[:, 0:3] = 1e6                                  # far outside the frustum
    k, q, d = conv.prepass(p, records=torch.from_numpy(rec).cuda())
    assert k == 0 and q.shape == (0, 24) and d.shape == (0,)
    assert oracle.prepass(p, rec)[0] == 0
    empty = torch.empty((0, 24), dtype=torch.float32, device="cuda")
    assert conv.prepass(p, records=empty)[0] == 0
    # errors: depth test requested without an image; resolution_target 0
    from dataclasses import replace
    from mesh2splat_amd._lib import M2SError
    with pytest.raises(M2SError):
        conv.prepass(replace(p, perform_mesh_depth_test=True), records=torch.from_numpy(rec).cuda())
    with pytest.raises(M2SError):
        conv.prepass(replace(p, resolution_target=0), records=torch.from_numpy(rec).cuda())


def test_prepass_is_deterministic_and_survives_chain_tag_wrap(conv, oracle):
    """Output order is input order on every run; the look-back chain's 16-bit tag wraps without harm (70 000 launches
    of a small input take a couple of seconds)."""
    import torch
    p = dict(CASES)["colour"]
    rec = prepass_cases.hostile_records(4096)
    d_rec = torch.from_numpy(rec).cuda()
    want = oracle.prepass(p, rec)
    first = conv.prepass(p, records=d_rec)
    assert_prepass_matches(first, want, 0, "first run")
    for i in range(70000):
        k = conv.prepass(p, records=d_rec, download=False)
        assert k == want[0], f"launch {i}"
    assert_prepass_matches(conv.prepass(p, records=d_rec), want, 0, "after the wrap")


def test_prepass_full_size_against_oracle(conv, oracle):
    """C3-scale input (cube-sphere n=289 at R=1024 -> 2.74 M records): count exact, floats bit-identical to the oracle."""
    scene = synth.cube_sphere(289, tex_size=256)
    conv.upload_scene(scene)
    conv.set_max_gaussians(0)
    total = conv.convert(1024)
    rec = conv.download()
    view, proj = prepass_cases.default_camera((1920, 1080))
    from mesh2splat_amd.prepass import PrepassParams
    p = PrepassParams(view_mat=view, proj_mat=proj, renderer_resolution=(1920, 1080), resolution_target=1024)
    conv.set_profiling(True)
    got = conv.prepass(p)
    conv.set_profiling(False)
    want = oracle.prepass(p, rec)
    assert_prepass_matches(got, want, 0, "C3")
    assert 0.3 * total < got[0] <= total
    assert 0.0 < conv.last_prepass_ms < 50.0


def _needles(n, seed=11):
    """Needle-shaped Gaussians in front of the default camera: one scale 10^4..10^6 times the others, random orientation.  Their
    screen-space covariance is numerically rank 1, so lambda2 = (mid - delta) / 2 is pure rounding noise and comes out negative
    for part of them — the shader's LAST cull test (gaussianSplattingPrepassCS.glsl:183) then removes them."""
    rng = np.random.default_rng(seed)
    r = np.zeros((n, 24), np.float32)
    r[:, 0:3] = rng.uniform(-0.6, 0.6, (n, 3))
    r[:, 3] = 1
    r[:, 4:8] = rng.uniform(0.2, 1, (n, 4))
    r[:, 8] = np.exp(rng.uniform(np.log(3e2), np.log(3e4), n))
    r[:, 9:11] = 1e-2
    q = rng.normal(size=(n, 4))
    r[:, 16:20] = q / np.linalg.norm(q, axis=1, keepdims=True)
    r[:, 12:15] = (0, 0, 1)
    r[:, 20:22] = 0.5
    r[:, 23] = 1
    return r


def test_the_last_cull_test_is_reproduced_on_needles(conv, oracle):
    """lambda2 < 0 (the shader's last test) only ever triggers through rounding; on 20 000 needle-shaped Gaussians, all inside the
    frustum, it removes thousands.  Which ones is a matter of single roundings in the covariance projection: count, order and every
    bit must be the oracle's."""
    import torch
    p = dict(CASES)[CASES[0][0]]
    ordinary = prepass_cases.base_records(oracle, 14, 64)
    rec = np.concatenate([ordinary, _needles(20000), ordinary[::-1]])
    want = oracle.prepass(p, rec)
    only_early = oracle.prepass(p, np.concatenate([ordinary, ordinary[::-1]]))
    assert only_early[0] + 10000 < want[0] < only_early[0] + 19000          # most needles survive, thousands fail the last test
    got = conv.prepass(p, records=torch.from_numpy(rec).cuda())
    assert_prepass_matches(got, want, p.render_mode, "needles")


@pytest.mark.parametrize("name", ["colour", "ply_classic", "inside", "depth_test", "model_trs"])
def test_prepass_sorted_is_prepass_then_radix_sort_in_one_pass(conv, oracle, name):
    """m2s_prepass_sorted: the frame's depth sort taken FIRST (permutation by the depth bits the prepass is going to store), the prepass
    through it — against m2s_prepass + m2s_sort_prepass on the same records (byte for byte) and against the oracle's prepass ordered by
    numpy's stable argsort.  Duplicated records (equal keys: stability), hostile records (NaN / huge / culled), a model matrix that is
    not the identity, a second frame from another camera (the position plane of the first is reused), new records (it is not).
    Without a depth image ("colour", "ply_classic", "model_trs"; "inside": a camera inside the object, most records behind the eye) the sort
    applies the frustum test itself and the prepass runs densely over the survivors; with one ("depth_test") the prepass compacts as in
    m2s_prepass."""
    from dataclasses import replace
    p = dict(CASES)[name]
    rec = np.concatenate([prepass_cases.base_records(oracle, 20, 96), prepass_cases.hostile_records(4096)])
    rec = np.concatenate([rec, rec[:5000]])
    if name == "ply_classic":
        rec = rec.copy()
    conv.upload_records(rec)
    for frame in range(2):
        if frame == 1:           # another camera and a model matrix with rotation + non-uniform scale + translation
            view, proj = prepass_cases.default_camera((800, 450))
            model = np.array([[0.0, 1.3, 0.0, 0.0], [-0.7, 0.0, 0.0, 0.0], [0.0, 0.0, 1.1, 0.0], [0.05, -0.1, 0.2, 1.0]], np.float32)   # column-major rows = columns
            p = replace(p, view_mat=view, proj_mat=proj, model_mat=model, renderer_resolution=(800, 450))
        wk, wq, wd = oracle.prepass(p, rec)
        want = wq[np.argsort(wd.view(np.uint32), kind="stable")]
        gk, _, _ = conv.prepass(p)
        two_calls = conv.sort_prepass()
        assert gk == wk and same_bits(two_calls, want).all()
        if name == "inside" and frame == 0:
            assert 0 < wk < rec.shape[0] // 2          # (most of the records are culled: the dense prepass runs over a minority)
        conv.set_profiling(True)
        fused = conv.prepass_sorted(p)
        conv.set_profiling(False)
        assert fused.shape == (wk, 24), (frame, fused.shape, wk)
        assert np.array_equal(bits(fused), bits(two_calls)), (frame, np.argwhere(bits(fused) != bits(two_calls))[:4].tolist())
        st = conv.last_sort_stage_ms          # keys | radix sort | (here) the prepass through the permutation
        assert st["keys"] > 0 and st["radix_sort"] > 0 and st["gather"] > 0
    # new records: the cached position plane belongs to the old ones and must not be used
    rec2 = rec[::-1].copy()
    rec2[:, 0:3] *= np.float32(0.5)
    conv.upload_records(rec2)
    wk, wq, wd = oracle.prepass(p, rec2)
    fused = conv.prepass_sorted(p)
    assert fused.shape == (wk, 24) and same_bits(fused, wq[np.argsort(wd.view(np.uint32), kind="stable")]).all()
    # nothing visible -> nothing sorted
    far = rec.copy()
    far[:, 0:3] = 1e6
    conv.upload_records(far)
    assert conv.prepass_sorted(p, download=False) == 0


def test_prepass_sorted_when_a_survivors_depth_bits_are_the_culled_marker(conv, oracle):
    """The sort marks culled records with the key 0xFFFFFFFF.  A record whose own depth IS that bit pattern (a NaN position with an all-ones
    payload is not culled: every comparison of the frustum test is false) makes the marker ambiguous: m2s_prepass_sorted then sorts
    everything and lets the prepass compact, as with a depth image.  Either way: the bytes of m2s_prepass + m2s_sort_prepass."""
    p = dict(CASES)["colour"]
    rec = prepass_cases.base_records(oracle, 12, 64).copy()
    allones = np.array([0xFFFFFFFF], np.uint32).view(np.float32)[0]
    rec[5::97, 0] = allones
    rec[7::131, 2] = allones
    conv.upload_records(rec)
    gk, _, _ = conv.prepass(p)
    two_calls = conv.sort_prepass()
    fused = conv.prepass_sorted(p)
    assert fused.shape == two_calls.shape == (gk, 24)
    assert np.array_equal(bits(fused), bits(two_calls))

"""-m gpu: round-2 device paths — conversions at densities the context has not seen (no count pass, band bases left behind
by the launch itself), the grow-only record pool, stale-record detection for overlapping submissions, the device-side
.ply row encoder (formats 1 and 2), slice export, and the RCCL exchange through the C ABI (world size 1)."""
import os

import numpy as np
import pytest

from mesh2splat_amd import dist as m2d
from mesh2splat_amd import synth
from mesh2splat_amd._lib import M2SError
from mesh2splat_amd.converter import Converter, write_ply
from mesh2splat_amd.scene import reference_cap
from parity import assert_records_match

pytestmark = pytest.mark.gpu


def test_new_densities_and_band_bases(hiplib, oracle):
    """> 196 608 triangles: 64-triangle batches, XCD bands.  The first launch at an R runs without bands and leaves the
    bases behind, the second reads them: same bytes, and the same as the multi-pass pipeline's."""
    scene = synth.cube_sphere(130, tex_size=64)          # 202 800 triangles
    c = Converter(0)
    c.upload_scene(scene)
    ref = Converter(0)
    ref.set_pipeline("multipass")
    ref.upload_scene(scene)
    for R in (640, 632, 648, 640):                        # never seen / seen
        want_total = ref.convert(R)
        want = ref.download()
        for rep in range(3):
            assert c.convert(R) == want_total
            assert c.last_pipeline == "lean"              # (k_fused3 and k_fused2 share units and run tables)
            assert np.array_equal(c.download().view(np.uint32), want.view(np.uint32)), (R, rep)
    ototal, orec, _ = oracle.convert(scene, 640, cap=reference_cap(640, 1))
    assert ototal == ref.convert(640)
    assert_records_match(ref.download(), orec, "new-R")
    # pipelined submissions at a new density: the first goes through the blocking path, the rest read the bands
    for _ in range(2):
        c.submit(656); c.submit(656); c.submit(656)
        t = [c.wait() for _ in range(3)]
        assert t[0] == t[1] == t[2] == ref.convert(656)
    assert np.array_equal(c.download().view(np.uint32), ref.download().view(np.uint32))
    c.close(); ref.close()


def test_record_pool_only_grows(hiplib, oracle):
    scene = synth.cube_sphere(12, tex_size=16)
    c = Converter(0)
    c.upload_scene(scene)
    ptrs = []
    for R in (64, 96, 64, 200, 96, 64):
        total = c.convert(R)
        ototal, orec, _ = oracle.convert(scene, R, cap=reference_cap(R, 1))
        assert total == ototal
        assert_records_match(c.download(), orec, f"R={R}")
        ptrs.append(c.device_records)
    assert ptrs[3] == ptrs[4] == ptrs[5]                  # after the largest cap: no re-allocation when R shrinks again
    c.close()


def test_unlimited_cap_sizes_from_the_prediction(hiplib, oracle):
    scene = synth.sphere_grid(2, n=4, tex_size=16)
    c = Converter(0)
    c.upload_scene(scene)
    c.set_max_gaussians(0)
    for R in (48, 300, 64, 512):                          # 40x more fragments than the first conversion predicted room for
        total = c.convert(R)
        ototal, orec, _ = oracle.convert(scene, R, cap=0)
        assert total == ototal == c.num_stored
        assert_records_match(c.download(), orec, f"R={R}")
    c.close()


def test_download_after_each_wait_with_mixed_densities(hiplib, oracle):
    """submit(128); submit(128); submit(64) into the context-owned buffer — a waited conversion whose records a
    later submission at another R has overwritten must not be handed out as if intact."""
    scene = synth.cube_sphere(24, tex_size=32)
    c = Converter(0)
    c.upload_scene(scene)
    want = {}
    for R in (128, 64):
        c.convert(R)
        want[R] = c.download().copy()
    c.convert(128)
    c.submit(128); c.submit(128); c.submit(64)
    assert c.wait() == len(want[128])
    with pytest.raises(M2SError):                          # overwritten by the R = 64 submission
        c.download()
    assert c.wait() == len(want[128])
    with pytest.raises(M2SError):
        c.download()
    assert c.wait() == len(want[64])
    assert np.array_equal(c.download().view(np.uint32), want[64].view(np.uint32))
    # same R in flight: the buffer holds exactly these records whichever conversion wrote last
    c.submit(64); c.submit(64)
    c.wait()
    assert np.array_equal(c.download().view(np.uint32), want[64].view(np.uint32))
    c.wait()
    # caller-owned buffers: every conversion keeps its own records
    import torch
    bufs = {R: torch.zeros((len(want[R]) + 8, 24), dtype=torch.float32, device="cuda") for R in (128, 64)}
    s = torch.cuda.current_stream().cuda_stream
    c.submit(128, bufs[128].data_ptr(), bufs[128].shape[0], s)
    c.submit(64, bufs[64].data_ptr(), bufs[64].shape[0], s)
    assert c.wait() == len(want[128]) and c.wait() == len(want[64])
    torch.cuda.synchronize()
    for R in (128, 64):
        got = bufs[R][: len(want[R])].cpu().numpy()
        assert np.array_equal(got.view(np.uint32), want[R].view(np.uint32))
    # two different caller streams: the context orders the second conversion behind the first (shared work buffers)
    s2 = torch.cuda.Stream()
    c.submit(128, bufs[128].data_ptr(), bufs[128].shape[0], s)
    c.submit(64, bufs[64].data_ptr(), bufs[64].shape[0], s2.cuda_stream)
    assert c.wait() == len(want[128]) and c.wait() == len(want[64])
    torch.cuda.synchronize()
    for R in (128, 64):
        assert np.array_equal(bufs[R][: len(want[R])].cpu().numpy().view(np.uint32), want[R].view(np.uint32))
    c.close()


@pytest.mark.parametrize("fmt", [1, 2])
def test_device_row_encoder_is_byte_identical(tmp_path, hiplib, fmt):
    """m2s_export_ply encodes formats 1 and 2 on the device (k_encode_rows, its own restatement of glibc's logf); the host
    writer m2s_write_ply is byte-identical to the reference's (tests/test_ref_host.py).  Same file, byte for byte."""
    scene = synth.cube_sphere(60, tex_size=128)          # 43 200 triangles, all three maps
    c = Converter(0)
    c.upload_scene(scene)
    for R, std in ((700, 0.65), (97, 1.3)):
        c.convert(R)
        rec = c.download()
        rec_special = rec.copy()
        dev = tmp_path / "dev.ply"
        host = tmp_path / "host.ply"
        c.export_ply(str(dev), fmt, std)
        write_ply(str(host), rec, fmt, np.float32(std) / np.float32(R))
        assert dev.read_bytes() == host.read_bytes(), (fmt, R)
        # special values through the same path: opaque / transparent alpha, zero and tiny scales, flipped normals
        rec_special[::5, 7] = 1.0; rec_special[1::5, 7] = 0.0; rec_special[2::5, 8] = 0.0; rec_special[3::5, 9] = 1e-42
        rec_special[::3, 12:15] *= -1.0
        c.upload_records(rec_special)
        # uploaded records carry no R: export must refuse, the explicit writer is the way
        with pytest.raises(M2SError):
            c.export_ply(str(dev), fmt, std)
    # the special values: put them into the context-owned records of a real conversion (test-only host -> device copy)
    import ctypes as C
    import torch
    c.convert(64)
    n = c.num_stored
    assert 0 < n <= len(rec_special)
    src = np.ascontiguousarray(rec_special[:n])
    rt = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    assert rt.hipMemcpy(C.c_void_p(c.device_records), C.c_void_p(src.ctypes.data), C.c_size_t(n * 96), C.c_int(1)) == 0
    dev = tmp_path / "dev2.ply"; host = tmp_path / "host2.ply"
    c.export_ply(str(dev), fmt, 0.65)
    write_ply(str(host), rec_special[:n], fmt, np.float32(0.65) / np.float32(64))
    assert dev.read_bytes() == host.read_bytes()
    c.close()


@pytest.mark.parametrize("fmt", [0, 2])
def test_slice_export_equals_whole_export(tmp_path, hiplib, fmt):
    scene = synth.cube_sphere(24, tex_size=32)
    c = Converter(0)
    c.upload_scene(scene)
    c.convert(256)
    n = c.num_stored
    whole = tmp_path / "w.ply"
    c.export_ply(str(whole), fmt, 0.65)
    # "two ranks": the same records written as rows [n, 2n) and [0, n) of a 2n-row file == header(2n) + body twice
    p = tmp_path / "s.ply"
    c.export_ply_slice(str(p), fmt, 0.65, n, n, 2 * n)
    c.export_ply_slice(str(p), fmt, 0.65, 0, n, 2 * n)
    w = whole.read_bytes()
    hdr_end = w.index(b"end_header\n") + len(b"end_header\n")
    s = p.read_bytes()
    shdr_end = s.index(b"end_header\n") + len(b"end_header\n")
    assert s[:shdr_end] == w[:hdr_end].replace(b"element vertex %d\n" % n, b"element vertex %d\n" % (2 * n))
    assert s[shdr_end:] == w[hdr_end:] * 2
    c.close()


def test_rccl_exchange_through_the_c_abi_world_1(hiplib, oracle):
    """One rank: communicator creation, counter exchange (blocking and pipelined) and the record 'gather' all run through
    librccl as opened by libm2s_hip.so (the N > 1 schedule is the same code with more peers; gloo-tested on CPU)."""
    import torch
    scene = synth.cube_sphere(16, tex_size=16)
    R = 128
    plan = m2d.shard_ranges_native(scene, R, 1)
    assert plan == [(0, scene.n_triangles)]
    ex = m2d.RcclExchange(0, 0, 1, lambda ident: ident)
    c = Converter(0)
    c.set_triangle_range(*plan[0])
    c.upload_scene(scene)
    c.set_max_gaussians(0)
    total = c.convert(R)
    counts, offs = ex.all_gather_counts(total)
    assert counts == [total] and offs == [0, total]
    for k in range(5):
        ex.publish_count(total + k)
    assert [ex.collect_counts()[0][0] for _ in range(5)] == [total + k for k in range(5)]
    with pytest.raises(M2SError):
        ex.collect_counts()
    merged = torch.zeros((total, 24), dtype=torch.float32, device="cuda")
    ex.gather_records(c.device_records, counts, merged.data_ptr(), -1, 0)
    torch.cuda.synchronize()
    assert np.array_equal(merged.cpu().numpy().view(np.uint32), c.download().view(np.uint32))
    ex.close(); c.close()


def test_set_records_adopts_device_memory(tmp_path, hiplib):
    """m2s_set_records: records that live in somebody else's device memory (the merged buffer of a multi-GPU exchange) become a
    context's current records without a copy — export, download and the depth sort then apply to them."""
    import torch
    scene = synth.cube_sphere(16, tex_size=32)
    a = Converter(0)
    a.upload_scene(scene)
    a.convert(160)
    rec = a.download()
    want = tmp_path / "want.ply"
    a.export_ply(str(want), 2, 0.65)
    merged = torch.from_numpy(rec.copy()).cuda()            # "somebody else's" buffer
    b = Converter(0)                                         # a context that never converted anything
    b.set_records(merged.data_ptr(), len(rec), 160)
    assert b.num_stored == len(rec)
    assert np.array_equal(b.download().view(np.uint32), rec.view(np.uint32))
    got = tmp_path / "got.ply"
    b.export_ply(str(got), 2, 0.65)
    assert got.read_bytes() == want.read_bytes()
    view = np.eye(4, dtype=np.float32)
    view[2, 3] = -3.0
    assert np.array_equal(b.sort_by_depth(view).view(np.uint32), a.sort_by_depth(view).view(np.uint32))
    assert b.sort_by_depth(view, download=False) == len(rec)
    a.close(); b.close()

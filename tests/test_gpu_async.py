"""-m gpu: m2s_convert_submit / m2s_convert_wait — back-to-back conversions without a host round trip per call."""
import numpy as np
import pytest

from mesh2splat_amd import synth
from mesh2splat_amd._lib import M2SError
from mesh2splat_amd.converter import Converter
from parity import assert_records_match

pytestmark = pytest.mark.gpu


def test_submit_wait_matches_synchronous(hiplib, oracle):
    scene = synth.cube_sphere(24, tex_size=64)
    R = 200
    a = Converter(0)
    a.upload_scene(scene)
    want_total = a.convert(R)
    want = a.download()
    ototal, orec, _ = oracle.convert(scene, R, cap=a_cap(scene, R))
    assert want_total == ototal
    assert_records_match(want, orec, "sync")

    b = Converter(0)
    b.upload_scene(scene)
    b.submit(R)                       # first conversion of (scene, R): executed inside submit
    assert b.wait() == want_total
    for depth in (1, 2, 4):           # then truly asynchronous, up to M2S_MAX_IN_FLIGHT deep
        for _ in range(3):
            for _ in range(depth):
                b.submit(R)
            for _ in range(depth):
                assert b.wait() == want_total
        assert b.num_stored == len(want)
        assert np.array_equal(b.download().view(np.uint32), want.view(np.uint32))
    a.close(); b.close()


def a_cap(scene, R):
    from mesh2splat_amd.scene import reference_cap
    return reference_cap(R, scene.n_meshes)


def test_submit_falls_back_for_deferred_triangles_and_multipass(hiplib, oracle):
    """Scenes that need the second stage (big triangles) or the multi-pass pipeline still give the right answer."""
    for scene, R in [(synth.unit_quad(), 256), (synth.sphere_grid(2, n=3, tex_size=16), 300)]:
        c = Converter(0)
        c.upload_scene(scene)
        ototal, orec, _ = oracle.convert(scene, R, cap=a_cap(scene, R))
        for _ in range(3):
            c.submit(R)
            c.submit(R)
            assert c.wait() == ototal and c.wait() == ototal
        assert_records_match(c.download(), orec, "fallback")
        c.close()
    c = Converter(0)
    c.set_pipeline("multipass")
    scene = synth.cube_sphere(10, tex_size=16)
    c.upload_scene(scene)
    ototal, orec, _ = oracle.convert(scene, 128, cap=a_cap(scene, 128))
    c.submit(128)
    assert c.wait() == ototal
    assert_records_match(c.download(), orec, "multipass via submit")
    first = c.download()
    for _ in range(3):               # from the second conversion on the four kernels are enqueued without waiting
        c.submit(128); c.submit(128); c.submit(128)
        assert c.wait() == ototal and c.wait() == ototal and c.wait() == ototal
    assert np.array_equal(c.download().view(np.uint32), first.view(np.uint32))
    c.close()
    # AUTO deciding for the multi-pass pipeline (mid-size triangles) pipelines the same way
    scene = synth.cube_sphere(20, tex_size=32)
    R = 512
    c = Converter(0)
    c.upload_scene(scene)
    ototal, orec, _ = oracle.convert(scene, R, cap=a_cap(scene, R))
    assert ototal > 11 * scene.n_triangles
    for _ in range(3):
        c.submit(R); c.submit(R)
        assert c.wait() == ototal and c.wait() == ototal
    assert_records_match(c.download(), orec, "AUTO -> multipass via submit")
    assert c.last_kernel_ms()["fused"] == 0.0
    c.close()


def test_submit_state_errors_and_resolution_change(hiplib, oracle):
    scene = synth.cube_sphere(12, tex_size=32)
    c = Converter(0)
    with pytest.raises(M2SError):
        c.wait()                                      # nothing in flight
    with pytest.raises(M2SError):
        c.submit(64)                                  # no scene
    c.upload_scene(scene)
    c.submit(64); assert c.wait() > 0                 # warm (sync inside)
    for _ in range(4):
        c.submit(64)
    with pytest.raises(M2SError):
        c.submit(64)                                  # ring full
    with pytest.raises(M2SError):
        c.convert(64)                                 # synchronous call while conversions are in flight
    with pytest.raises(M2SError):
        c.upload_scene(scene)
    totals = [c.wait() for _ in range(4)]
    assert len(set(totals)) == 1
    # a different R goes through the synchronous path again (buffer re-sized by the cap formula), then async
    o128, _, _ = oracle.convert(scene, 128, cap=a_cap(scene, 128))
    c.submit(128); c.submit(128); c.submit(64)
    assert c.wait() == o128 and c.wait() == o128 and c.wait() == totals[0]
    c.close()


def test_submit_into_user_buffer_on_user_stream(hiplib, oracle):
    torch = pytest.importorskip("torch")
    scene = synth.cube_sphere(16, tex_size=32)
    R = 160
    ototal, orec, _ = oracle.convert(scene, R, cap=0)
    c = Converter(0)
    c.upload_scene(scene)
    c.set_max_gaussians(0)
    bufs = [torch.zeros((ototal, 24), dtype=torch.float32, device="cuda") for _ in range(2)]
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for i in range(6):                            # alternate two buffers, two in flight
            c.submit(R, bufs[i % 2].data_ptr(), ototal, st.cuda_stream)
            if i:
                assert c.wait() == ototal
        assert c.wait() == ototal
    st.synchronize()
    for b in bufs:
        assert_records_match(b.cpu().numpy(), orec, "user buffer")
    c.close()


def test_two_lanes_overlap_gives_the_same_records(hiplib, oracle):
    """m2s_set_async_lanes(2): consecutive context-owned conversions run on two streams with their own chains and record
    buffers and overlap; every wait returns the right counter and the records last waited for are complete and identical."""
    scene = synth.cube_sphere(100, tex_size=64)       # 120 000 triangles, single-pass kernel
    R = 384
    c = Converter(0)
    c.upload_scene(scene)
    want_total = c.convert(R)
    want = c.download()
    ototal, orec, _ = oracle.convert(scene, R, cap=a_cap(scene, R))
    assert want_total == ototal
    assert_records_match(want, orec, "blocking")
    c.set_async_lanes(2)
    for depth in (2, 3, 4):
        for _ in range(depth):
            c.submit(R)
        for i in range(12):
            assert c.wait() == want_total
            # the buffer of the conversion just waited for is complete, whichever lane it ran on
            if i % 5 == 0:
                assert np.array_equal(c.download().view(np.uint32), want.view(np.uint32))
            c.submit(R)
        for _ in range(depth):
            assert c.wait() == want_total
        assert np.array_equal(c.download().view(np.uint32), want.view(np.uint32))
    with pytest.raises(M2SError):
        c.submit(R); c.set_async_lanes(1)             # not while conversions are in flight
    c.wait()
    c.set_async_lanes(1)
    c.submit(R)
    assert c.wait() == want_total
    c.close()


@pytest.mark.parametrize("forced", [True, False])
def test_two_lanes_multipass_gives_the_same_records(hiplib, oracle, forced):
    """The multi-pass pipeline on two lanes: the second lane has its own offsets / slice starts / TriSetup records / counter, so
    k_count_scan of one conversion runs beside k_emit2 of the one before — same counters, same bytes, whichever lane the conversion
    last waited for ran on; a change of density in between re-sizes the second lane's tables."""
    scene = synth.cube_sphere(40, tex_size=64)         # 19 200 triangles; at R = 640 about 50 fragments each: AUTO takes the multi-pass pipeline
    c = Converter(0)
    if forced:
        c.set_pipeline("multipass")
    c.upload_scene(scene)
    c.set_max_gaussians(0)
    wants = {}
    for R in (640, 900):
        total = c.convert(R)
        assert c.last_pipeline == "multipass"
        rec = c.download()
        ototal, orec, _ = oracle.convert(scene, R, cap=0)
        assert total == ototal
        assert_records_match(rec, orec, f"blocking R={R}")
        wants[R] = (total, rec)
    c.set_async_lanes(2)
    for R in (640, 900, 640):
        want_total, want = wants[R]
        for depth in (2, 3, 4):
            for _ in range(depth):
                c.submit(R)
            for i in range(9):
                assert c.wait() == want_total
                if i % 4 == 0:
                    assert np.array_equal(c.download().view(np.uint32), want.view(np.uint32))
                c.submit(R)
            for _ in range(depth):
                assert c.wait() == want_total
            assert np.array_equal(c.download().view(np.uint32), want.view(np.uint32))
    # alternating densities: every conversion waited for has its own counter, and the records follow it
    seq = [640, 900, 900, 640, 640, 900]
    for R in seq[:3]:
        c.submit(R)
    for i, R in enumerate(seq):
        assert c.wait() == wants[R][0]
        if i + 3 < len(seq):
            c.submit(seq[i + 3])
    c.set_async_lanes(1)
    assert c.convert(640) == wants[640][0]
    assert np.array_equal(c.download().view(np.uint32), wants[640][1].view(np.uint32))
    c.close()


def test_two_lanes_with_extra_blocks_by_ticket(hiplib):
    """The heterogeneous scene has 1043 blocks of 256 triangles for 1024 resident workgroups: k_count_scan launches the resident set and
    hands 19 extra blocks out by ticket (m2s_emit2.hip).  On two lanes its workgroups share the GPU with the other lane's k_emit2 — not
    all of a launch is resident at once, and a workgroup with an extra block waits for workgroups dispatched after it: sixty overlapped
    conversions must all deliver the blocking conversion's counter and bytes (a stalled look-back would surface as an error)."""
    scene = synth.sponza_like(tex_scale=0.25)
    c = Converter(0)
    c.upload_scene(scene)
    R = 1024
    want_total = c.convert(R)
    assert c.last_pipeline == "multipass" and c.num_triangles > 1024 * 256
    want = c.download()
    c.set_async_lanes(2)
    for depth in (2, 4):
        for _ in range(depth):
            c.submit(R)
        for i in range(30):
            assert c.wait() == want_total
            if i % 10 == 0:
                assert np.array_equal(c.download().view(np.uint32), want.view(np.uint32))
            c.submit(R)
        for _ in range(depth):
            assert c.wait() == want_total
        assert np.array_equal(c.download().view(np.uint32), want.view(np.uint32))
    c.set_async_lanes(1)
    c.close()

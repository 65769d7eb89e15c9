"""The entry point that allocates nothing (lpt_render_frame_static, include/lpt.h) against the entry point it
is a variant of (lpt_render_frame), through the C-ABI: same bits, on calls in a row that share the static
scratch array, the later ones with a smaller image."""
import numpy as np
import pytest

import logipathtracer_amd as L

pytestmark = pytest.mark.gpu


def _frames(dev, ubos, static):
    for i in range(len(ubos)):
        if static:
            dev.render_frame_static(ubos[i:i + 1])
        else:
            dev.render_frames(ubos[i:i + 1])
    img = dev.read_accum()
    return img, dev.stats()


def test_static_entry_point_equals_render_frame(cornell):
    bounces = 8
    ubos = L.frame_ubos(cornell.camera_world, cornell.fov_y, 3, seed=0x5EED)
    a, b = L.Device(0), L.Device(0)
    try:
        for dev in (a, b):
            dev.upload_scene(cornell)
            dev.set_options(bounces, L.BSDF_HEITZ, L.PIPELINE_AUTO, 0)
        # two renders in a row through the shared scratch, the second with a smaller image; then a larger one again
        for (w, h) in ((320, 200), (96, 56), (203, 77)):
            a.resize(w, h)
            b.resize(w, h)
            want, wst = _frames(a, ubos, static=False)
            got, gst = _frames(b, ubos, static=True)
            assert gst.paths == wst.paths == w * h and gst.segments == wst.segments
            assert np.all(got[..., 3] == 3.0)
            assert np.array_equal(got, want), (w, h)
        # the calls leave the amount of free device memory where it was
        torch = pytest.importorskip("torch")
        b.synchronize()
        free0 = torch.cuda.mem_get_info(0)[0]
        for i in range(3):
            b.render_frame_static(ubos[i:i + 1])
        b.synchronize()
        assert torch.cuda.mem_get_info(0)[0] == free0
        # a sharded context works the same way
        for dev in (a, b):
            dev.resize(200, 120)
            dev.set_partition(32, 32, 1, 3)
        want, _ = _frames(a, ubos, static=False)
        got, _ = _frames(b, ubos, static=True)
        assert np.array_equal(got, want) and (got[..., 3] > 0).any()
        # two contexts on one GPU, each with a stream of its own, share the array: their calls are put in order
        # (hipStreamWaitEvent on the event of the last user), so frames enqueued back to back without any host-side
        # wait in between come out as if each context had the array to itself
        for dev in (a, b):
            dev.resize(640, 360)  # large enough for the launches of the two contexts to overlap if nothing ordered them
        want, wst = _frames(a, ubos, static=False)
        for dev in (a, b):
            dev.resize(640, 360)  # clears the images
        for i in range(len(ubos)):
            a.render_frame_static(ubos[i:i + 1])
            b.render_frame_static(ubos[i:i + 1])
        got_a, got_b = a.read_accum(), b.read_accum()
        assert np.array_equal(got_a, want) and np.array_equal(got_b, want)
        st = b.stats()
        assert st.paths == 640 * 360 and (st.timed_bounce_lo, st.timed_bounce_hi) == (0, bounces)
        # beyond the static array's size the call refuses instead of allocating
        b.resize(2048, 1160)
        with pytest.raises(L.LptError, match="static") as e:
            b.render_frame_static(ubos[:1])
        assert e.value.status == -2
        # the largest supported image (1920x1080) fits
        a.resize(1920, 1080)
        b.resize(1920, 1080)
        want, _ = _frames(a, ubos[:1], static=False)
        got, _ = _frames(b, ubos[:1], static=True)
        assert np.array_equal(got, want)
    finally:
        a.close()
        b.close()

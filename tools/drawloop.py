"""Times a plain RendererPT::drawFrame loop (one frame per call, no deferred submission) at three sizes.
   python tools/drawloop.py   (on a GPU box, from the repository root)"""
import sys, time
sys.path.insert(0, '.')
import logipathtracer_amd as L
for (w, h) in ((256, 256), (1280, 720), (1920, 1080)):
    r = L.Renderer(w, h)
    r.load_scene('tests/golden/cornell_box.gltf')
    r.configure(seed=1, max_bounces=10)
    r.draw_frames(20); r.read_accumulation()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        r.draw_frame()
    img = r.read_accumulation()
    dt = time.perf_counter() - t0
    print('%dx%d: %.3f ms per drawFrame (%d frames, alpha %.0f)' % (w, h, dt / n * 1e3, n, img[0, 0, 3]))
    r.close()

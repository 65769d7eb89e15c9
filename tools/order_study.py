"""Drives tools/order_study.c: traversal steps per segment under the shader's fixed child order, a
nearer-child-first order and the fixed order started from a known bound (CPU only, analysis aid).

    python tools/order_study.py [cornell|outside] [width height frames]
"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import logipathtracer_amd as L  # noqa: E402
from logipathtracer_amd import scenes  # noqa: E402


class SceneC(C.Structure):
    _fields_ = [("objects", C.c_void_p), ("nObjects", C.c_uint32), ("tlas", C.c_void_p), ("nTlas", C.c_uint32),
                ("vertices", C.c_void_p), ("nVertices", C.c_uint32), ("blas", C.c_void_p), ("nBlas", C.c_uint32)]


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "cornell"
    w, h, frames = (int(a) for a in sys.argv[2:5]) if len(sys.argv) > 4 else (240, 135, 2)
    rel = float(os.environ.get("REL", "0.01"))
    abs_ = float(os.environ.get("ABS", "0.001"))
    so = "/tmp/liborder_study.so"
    subprocess.run(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", "-shared", "-fPIC", "-o", so,
                    os.path.join(ROOT, "tools", "order_study.c"), "-lm"], check=True)
    lib = C.CDLL(so)
    if which == "cornell":
        path = os.path.join(ROOT, "tests", "golden", "cornell_box.gltf")
    else:
        path = os.path.join(tempfile.mkdtemp(), "outside_synth.gltf")
        scenes.write_outside_synth(path, float(os.environ.get("DETAIL", "1.0")))
    s = L.Scene(path)
    sc = SceneC(s.objects.ctypes.data, len(s.objects), s.object_bvh.ctypes.data, len(s.object_bvh),
                s.vertices.ctypes.data, len(s.vertices), s.mesh_bvh.ctypes.data, len(s.mesh_bvh))
    out = np.zeros(16 * 12, dtype=np.float64)
    cam = np.ascontiguousarray(s.camera_world, dtype=np.float32)
    lib.order_study(C.byref(sc), C.c_void_p(cam.ctypes.data), C.c_float(s.fov_y), w, h, frames, 8, C.c_float(rel),
                    C.c_float(abs_), C.c_void_p(out.ctypes.data))
    out = out.reshape(16, 12)
    print(f"{which}: {len(s.objects)} objects, {len(s.vertices) // 3} triangles; window rel={rel} abs={abs_}")
    print("bounce  segments | SH inner  tris  objs | NF inner  tris  objs | BD inner  tris  objs | ambiguous  mismatch")
    tot = out.sum(axis=0)
    for b, r in list(enumerate(out)) + [("1-7", out[1:8].sum(axis=0)), ("all", tot)]:
        if r[0] == 0:
            continue
        n = r[0]
        print(f"{b!s:>6} {int(n):9d} | {r[1]/n:8.2f} {r[2]/n:5.2f} {r[3]/n:5.2f} | {r[4]/n:8.2f} {r[5]/n:5.2f} {r[6]/n:5.2f} |"
              f" {r[7]/n:8.2f} {r[8]/n:5.2f} {r[9]/n:5.2f} | {r[10]/n:8.4f}  {r[11]:.0f}")


if __name__ == "__main__":
    main()

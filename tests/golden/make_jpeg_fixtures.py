"""Writes the JPEG fixtures of tests/test_textures.py: small files encoded by Pillow (libjpeg-turbo) and the
pixels Pillow itself decodes from them.  Run once where Pillow is installed:  python tests/golden/make_jpeg_fixtures.py
The files are data: the loader under test never sees Pillow."""
import io
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))


def picture(w, h, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.zeros((h, w, 3), np.float64)
    img[..., 0] = 255.0 * xx / max(w - 1, 1)
    img[..., 1] = 255.0 * yy / max(h - 1, 1)
    img[..., 2] = 128.0 + 100.0 * np.sin(xx / 3.0) * np.cos(yy / 5.0)
    img[h // 4:h // 2, w // 4:w // 2] = [250, 20, 30]  # a hard-edged patch
    img += rng.normal(0.0, 6.0, img.shape)
    return np.clip(np.round(img), 0, 255).astype(np.uint8)


def main():
    cases = {
        "jpeg_444": dict(size=(37, 29), mode="RGB", kw=dict(quality=92, subsampling=0)),
        "jpeg_422": dict(size=(50, 23), mode="RGB", kw=dict(quality=85, subsampling=1)),
        "jpeg_420_restart": dict(size=(67, 45), mode="RGB", kw=dict(quality=90, subsampling=2, restart_marker_blocks=3)),
        "jpeg_grey": dict(size=(33, 18), mode="L", kw=dict(quality=88)),
        "jpeg_optimized": dict(size=(40, 40), mode="RGB", kw=dict(quality=75, subsampling=2, optimize=True)),
        "jpeg_progressive": dict(size=(24, 24), mode="RGB", kw=dict(quality=80, progressive=True)),
        "jpeg_progressive_444": dict(size=(45, 31), mode="RGB", kw=dict(quality=93, progressive=True, subsampling=0,
                                                                        restart_marker_blocks=4)),
        "jpeg_progressive_grey": dict(size=(19, 26), mode="L", kw=dict(quality=70, progressive=True, optimize=True)),
        "jpeg_cmyk": dict(size=(16, 16), mode="CMYK", kw=dict(quality=80)),
    }
    expected = {}
    for i, (name, c) in enumerate(cases.items()):
        w, h = c["size"]
        px = picture(w, h, 100 + i)
        im = Image.fromarray(px).convert(c["mode"])
        buf = io.BytesIO()
        im.save(buf, format="JPEG", **c["kw"])
        data = buf.getvalue()
        with open(os.path.join(HERE, name + ".jpg"), "wb") as f:
            f.write(data)
        expected[name] = np.array(Image.open(io.BytesIO(data)).convert("RGBA"))
        print(name, len(data), "bytes", expected[name].shape)
    np.savez_compressed(os.path.join(HERE, "jpeg_expected.npz"), **expected)


if __name__ == "__main__":
    main()

"""Struct layouts of the interchange buffers and the exported C-ABI surface (no GPU needed)."""
import ctypes as C
import os
import re

import numpy as np

from conftest import ROOT
from logipathtracer_amd import _native as N
from oracle import binding as O


def _offsets(dt):
    return {n: dt.fields[n][1] for n in dt.names}


def test_object_layout_matches_reference_struct():
    # include/PTSceneConverter.hpp:15-42 / shaders/path_tracing.comp:41-57 (std430)
    o = _offsets(N.OBJECT_DTYPE)
    assert N.OBJECT_DTYPE.itemsize == 208
    assert o == {"worldMatrix": 0, "worldMatrixInverse": 64, "baseColorFactor": 128, "emissionFactor": 144,
                 "metallicFactor": 156, "roughnessFactor": 160, "transmissionFactor": 164, "ior": 168,
                 "colorTexture": 172, "emissionTexture": 176, "metallicRoughnessTexture": 180,
                 "transmissionTexture": 184, "normalTexture": 188, "bvhOffset": 192, "verticesOffset": 196,
                 "padding": 200}


def test_vertex_node_ubo_layouts():
    v, n, u = _offsets(N.VERTEX_DTYPE), _offsets(N.NODE_DTYPE), _offsets(N.UBO_DTYPE)
    assert N.VERTEX_DTYPE.itemsize == 48 and (v["position"], v["normal"], v["uv"]) == (0, 16, 32)
    assert N.NODE_DTYPE.itemsize == 48 and (n["minCorner"], n["maxCorner"], n["isLeaf"], n["indices"]) == (0, 16, 28, 32)
    # include/RendererPT.h:51-61: mat4 0, fovY 64, pad to 80, uvec2 80, reset 88 -> 92 bytes
    assert N.UBO_DTYPE.itemsize == 92 and (u["fovY"], u["seed"], u["reset"]) == (64, 80, 88)


def test_oracle_structs_have_the_same_sizes():
    assert C.sizeof(O.Scene) == 80 and C.sizeof(O.Texture) == 32
    assert C.sizeof(O.Params) == 36


def test_every_declared_entry_point_is_exported():
    text = open(os.path.join(ROOT, "include", "lpt.h")).read()
    declared = set(re.findall(r"^\s*(?:int|void|const char\*)\s+(lpt_[a-z_0-9]+)\s*\(", text, flags=re.M))
    assert declared == set(N.LPT_SYMBOLS), declared ^ set(N.LPT_SYMBOLS)
    for name in declared:
        assert getattr(N.hip, name) is not None


def test_device_layer_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        return
    import logipathtracer_amd as L
    try:
        L.Device(0)
    except L.LptError as e:
        assert e.status == -4 and "HIP device" in str(e)
    else:
        raise AssertionError("lpt_create succeeded without a GPU")
    try:
        L.Renderer(64, 64)
    except L.LptError as e:
        assert "device context" in str(e)
    else:
        raise AssertionError("RendererPT was created without a GPU")

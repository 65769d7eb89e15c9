"""Physics pins for the CPU oracle.

The reference ships no tests or golden vectors, so the oracle's known-answer tests are derived from the
same shader text the oracle restates.  The checks here do not depend on reading that text: they are
properties any correct implementation of the path tracer must have — white-furnace energy conservation
per lobe, the Fresnel reflect/refract split and Snell's law, the mirror and Lambert limits of the
microfacet BSDFs, and the statistics of the Monte Carlo estimator (independent runs agree within their
error, the error shrinks as 1/sqrt(N))."""
import ctypes as C

import numpy as np
import pytest

import logipathtracer_amd as L
from logipathtracer_amd import scenes
from oracle import binding as O

F32 = np.float32


def _flat_sphere(level=3):
    """Icosphere with per-face normals: with geometric normals a direction in the upper hemisphere of the
    shading frame always leaves a convex body."""
    v, idx = scenes._icosphere(level)
    tri = v[idx.reshape(-1, 3)]                      # (n, 3, 3)
    n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    n = np.where((n * tri.mean(axis=1)).sum(axis=1, keepdims=True) < 0, -n, n)  # outward
    pos = tri.reshape(-1, 3).astype(np.float32)
    nrm = np.repeat(n, 3, axis=0).astype(np.float32)
    return pos, nrm, np.arange(len(pos), dtype=np.uint32)


def _sphere_scene(tmp_path, name, **material):
    g = scenes._GltfBuilder()
    m = g.material("M", (1.0, 1.0, 1.0, 1.0), **material)
    pos, nrm, idx = _flat_sphere()
    g.node("Sphere", mesh=g.mesh("Sphere", [(pos, nrm, idx, m)]))
    g.node("Camera", camera=g.camera(0.6), translation=(0, 0, 4))
    path = str(tmp_path / (name + ".gltf"))
    g.write(path)
    s = L.Scene(path, builder=L.BUILDER_BINNED_SAH)
    return s, O.OracleScene(s.objects, s.object_bvh, s.vertices, s.mesh_bvh)


# ---- white furnace: a white, non-emitting body in the uniform 0.2 environment (the miss term) must look
# ---- exactly like the environment, whatever the lobe, as long as no energy is created or lost -------------

@pytest.mark.parametrize("roughness", [0.05, 0.5, 1.0])
def test_white_furnace_conductor(tmp_path, roughness):
    s, osc = _sphere_scene(tmp_path, "conductor", metallic=1.0, roughness=roughness)
    frames = 3
    img, st = O.render_frames(osc, L.frame_ubos(s.camera_world, s.fov_y, frames, seed=41), 64, 64, 32)
    assert st["hits"] > 0.2 * st["paths"]  # the sphere covers a good part of the image
    expect = F32(0.2) + F32(0.2) + F32(0.2)
    # every path leaves the convex body with throughput exactly 1 and picks up the environment
    assert np.all(img[..., :3] == expect)


def test_white_furnace_dielectric(tmp_path):
    s, osc = _sphere_scene(tmp_path, "glass", metallic=0.0, roughness=0.2, transmission=1.0, ior=1.5)
    frames = 2
    img, st = O.render_frames(osc, L.frame_ubos(s.camera_world, s.fov_y, frames, seed=43), 64, 64, 256)
    assert st["segments"] > 2.5 * st["paths"] * (st["hits"] / st["segments"])  # paths do travel through the body
    expect = F32(0.2) + F32(0.2)
    ok = np.all(img[..., :3] == expect, axis=-1)
    # light is neither absorbed nor created; at most a path trapped by total internal reflection for 256 bounces is lost
    assert ok.mean() >= 0.9995
    assert np.all(img[..., :3] <= expect)


@pytest.mark.parametrize("roughness,max_loss", [(0.1, 0.002), (0.5, 0.01), (1.0, 0.05)])
def test_white_furnace_diffuse_loses_only_the_killed_walks(tmp_path, roughness, max_loss):
    s, osc = _sphere_scene(tmp_path, "diffuse", metallic=0.0, roughness=roughness)
    frames = 4
    img, st = O.render_frames(osc, L.frame_ubos(s.camera_world, s.fov_y, frames, seed=47), 64, 64, 32)
    per_frame = img[..., 0] / F32(0.2)
    assert np.all(img[..., 0] == img[..., 1]) and np.all(img[..., 1] == img[..., 2])  # white stays white
    assert np.all(per_frame <= frames + 1e-5)                                          # nothing is created
    # only walks cut off at the 16th scattering order lose their energy
    loss = 1.0 - img[..., 0].astype(np.float64).mean() / (float(F32(0.2)) * frames)
    assert -1e-6 <= loss <= max_loss, loss


# ---- the dielectric interface ---------------------------------------------------------------------------

def _sample_bsdf(kind, view, roughness, ior, n, base=(1.0, 1.0, 1.0), outside=1):
    F0, v = O.v3(base), O.v3(view)
    out = np.empty((n, 3), dtype=np.float64)
    wts = np.empty((n, 3), dtype=np.float64)
    ld, w = (C.c_float * 3)(), (C.c_float * 3)()
    for i in range(n):
        s = O.Seed((i * 2654435761 + 12345) & 0xFFFFFFFF, (i * 40503 + 977) & 0xFFFFFFFF)
        O.lib.lpto_bsdf(O.BSDF_HEITZ, kind, F0, v, roughness, 1.0 if kind == O.K_TRANS else 0.0, ior, outside, C.byref(s), ld, w)
        out[i] = ld[0], ld[1], ld[2]
        wts[i] = w[0], w[1], w[2]
    return out, wts


def _fresnel_unpolarised(cos_i, n1, n2):
    sin_t = n1 / n2 * np.sqrt(max(0.0, 1 - cos_i * cos_i))
    if sin_t >= 1:
        return 1.0
    cos_t = np.sqrt(1 - sin_t * sin_t)
    rs = (n1 * cos_i - n2 * cos_t) / (n1 * cos_i + n2 * cos_t)
    rp = (n1 * cos_t - n2 * cos_i) / (n1 * cos_t + n2 * cos_i)
    return 0.5 * (rs * rs + rp * rp)


@pytest.mark.parametrize("theta_deg,ior", [(0.0, 1.5), (60.0, 1.5), (0.0, 1.8), (75.0, 1.33)])
def test_smooth_dielectric_splits_light_by_fresnel_and_refracts_by_snell(theta_deg, ior):
    th = np.radians(theta_deg)
    view = (np.sin(th), 0.0, np.cos(th))
    n = 20000
    dirs, wts = _sample_bsdf(O.K_TRANS, view, 0.001, ior, n)
    assert np.allclose(np.linalg.norm(dirs, axis=1), 1.0, atol=1e-4)
    reflected = dirs[:, 2] > 0
    F = _fresnel_unpolarised(np.cos(th), 1.0, ior)
    sigma = np.sqrt(F * (1 - F) / n)
    assert abs(reflected.mean() - F) <= 4.5 * sigma + 1e-4, (reflected.mean(), F)
    # the oracle's own Fresnel term agrees with the textbook formula
    assert abs(O.lib.lpto_fresnel(float(np.cos(th)), float(ior)) - F) < 1e-5
    # reflected light leaves in the mirror direction, transmitted light obeys Snell's law
    assert np.allclose(dirs[reflected], [-view[0], -view[1], view[2]], atol=5e-3)
    sin_t = np.sin(th) / ior
    want = np.array([-sin_t, 0.0, -np.sqrt(1 - sin_t * sin_t)])
    assert np.allclose(dirs[~reflected], want, atol=5e-3)
    assert np.all(wts == 1.0)  # a clear dielectric neither absorbs nor amplifies


def test_total_internal_reflection_from_inside():
    # inside glass (ior 1.5) the critical angle is 41.8 degrees: at 60 degrees nothing gets out
    th = np.radians(60.0)
    dirs, _ = _sample_bsdf(O.K_TRANS, (np.sin(th), 0.0, np.cos(th)), 0.001, 1.5, 2000, outside=0)
    assert np.all(dirs[:, 2] > 0)
    # at 20 degrees most of it does, by the Fresnel factor of the inside-out interface
    th = np.radians(20.0)
    dirs, _ = _sample_bsdf(O.K_TRANS, (np.sin(th), 0.0, np.cos(th)), 0.001, 1.5, 20000, outside=0)
    F = _fresnel_unpolarised(np.cos(th), 1.5, 1.0)
    assert abs((dirs[:, 2] > 0).mean() - F) <= 4.5 * np.sqrt(F * (1 - F) / 20000) + 1e-4
    out = dirs[dirs[:, 2] < 0]
    assert np.allclose(np.hypot(out[:, 0], out[:, 1]), 1.5 * np.sin(th), atol=5e-3)  # sin(theta_t) = n sin(theta_i)


# ---- limits of the microfacet lobes -----------------------------------------------------------------------

def test_smooth_conductor_is_a_mirror_and_keeps_its_colour():
    th = np.radians(35.0)
    view = (np.sin(th) * 0.6, np.sin(th) * 0.8, np.cos(th))
    dirs, wts = _sample_bsdf(O.K_METALLIC, view, 0.001, 1.5, 2000, base=(0.9, 0.5, 0.2))
    assert np.allclose(dirs, [-view[0], -view[1], view[2]], atol=5e-3)
    assert np.allclose(wts, [0.9, 0.5, 0.2], atol=1e-6)  # single scattering: one factor of F0


def test_rough_conductor_conserves_energy_on_white_and_darkens_with_order_on_colour():
    dirs, wts = _sample_bsdf(O.K_METALLIC, (0.6, 0.0, 0.8), 0.8, 1.5, 5000)
    assert np.all(wts == 1.0) and np.all(dirs[:, 2] > 0)   # white metal: every walk leaves with all its energy
    _, wts = _sample_bsdf(O.K_METALLIC, (0.6, 0.0, 0.8), 0.8, 1.5, 5000, base=(0.5, 0.5, 0.5))
    orders = np.round(np.log(wts[:, 0]) / np.log(0.5))
    assert np.allclose(wts[:, 0], 0.5 ** orders, rtol=1e-5) and orders.min() == 1 and orders.max() >= 3
    # multiple scattering is the minority but not negligible on a rough surface seen at 37 degrees
    assert 0.05 < (orders > 1).mean() < 0.6


def _ggx_lambda(w, alpha):
    """Smith's Lambda of the GGX distribution for unit vectors w (..., 3) with w.z > 0."""
    t2 = (w[..., 0] ** 2 + w[..., 1] ** 2) / np.maximum(w[..., 2] ** 2, 1e-300)
    return 0.5 * (np.sqrt(1.0 + alpha * alpha * t2) - 1.0)


@pytest.mark.parametrize("roughness,theta_deg", [(0.7, 50.0), (0.45, 20.0), (0.9, 70.0)])
def test_first_scattering_order_of_the_rough_conductor_is_the_smith_ggx_brdf(roughness, theta_deg):
    """Literature pin, independent of the shader's text: walks that leave the microsurface after ONE reflection are
    distributed as the classical single-scattering microfacet model with the GGX distribution and Smith's
    height-correlated masking-shadowing (Walter et al. 2007; Heitz 2014):
        p1(wo) = D(h) / (4 cos(theta_i) (1 + Lambda(wi) + Lambda(wo))),   h = (wi + wo)/|wi + wo|,  alpha = roughness^2.
    Orders are told apart by the weight (0.5 per reflection).  Binned over the hemisphere, 4.5 sigma per bin."""
    alpha = roughness * roughness
    th = np.radians(theta_deg)
    wi = np.array([np.sin(th), 0.0, np.cos(th)])
    n = 120000
    dirs, wts = _sample_bsdf(O.K_METALLIC, tuple(wi), roughness, 1.5, n, base=(0.5, 0.5, 0.5))
    order = np.round(np.log(wts[:, 0]) / np.log(0.5)).astype(int)
    first = (order == 1) & (dirs[:, 2] > 0)
    nz, nphi, sub = 6, 8, 48
    zi = np.minimum((dirs[first, 2] * nz).astype(int), nz - 1)
    ph = np.mod(np.arctan2(dirs[first, 1], dirs[first, 0]), 2 * np.pi)
    pi_ = np.minimum((ph / (2 * np.pi) * nphi).astype(int), nphi - 1)
    counts = np.zeros((nz, nphi))
    np.add.at(counts, (zi, pi_), 1.0)
    # the model, integrated over each bin (the solid-angle element is d(cos theta) d(phi): midpoint rule)
    z = (np.arange(nz * sub) + 0.5) / (nz * sub)
    phi = (np.arange(nphi * sub) + 0.5) / (nphi * sub) * 2 * np.pi
    Z, P = np.meshgrid(z, phi, indexing="ij")
    S = np.sqrt(1.0 - Z * Z)
    wo = np.stack([S * np.cos(P), S * np.sin(P), Z], axis=-1)
    h = wo + wi
    h /= np.linalg.norm(h, axis=-1, keepdims=True)
    D = alpha * alpha / (np.pi * ((alpha * alpha - 1.0) * h[..., 2] ** 2 + 1.0) ** 2)
    p1 = D / (4.0 * wi[2] * (1.0 + _ggx_lambda(wi, alpha) + _ggx_lambda(wo, alpha)))
    cell = (1.0 / (nz * sub)) * (2 * np.pi / (nphi * sub))
    model = (p1 * cell).reshape(nz, sub, nphi, sub).sum(axis=(1, 3))
    sigma = np.sqrt(model * (1.0 - model) / n)
    err = np.abs(counts / n - model)
    assert np.all(err <= 4.5 * sigma + 3e-4), (float((err / (sigma + 1e-12)).max()), counts.sum() / n, model.sum())
    # and so is their share of all walks: the single-scattering albedo of the model
    assert abs(first.mean() - model.sum()) <= 4.5 * np.sqrt(model.sum() * (1 - model.sum()) / n) + 1e-3


@pytest.mark.parametrize("alpha,theta_deg", [(0.5, 60.0), (0.25, 30.0), (0.9, 85.0)])
def test_visible_normals_follow_the_ggx_vndf(alpha, theta_deg):
    """Normals drawn for a view direction wi are distributed as the distribution of VISIBLE normals (Heitz 2014):
    D_wi(m) = G1(wi) max(0, wi.m) D(m) / cos(theta_i), G1 = 1 / (1 + Lambda(wi)) — which integrates to one."""
    th = np.radians(theta_deg)
    wi = np.array([np.sin(th), 0.0, np.cos(th)])
    n = 60000
    out = (C.c_float * 3)()
    ms = np.empty((n, 3))
    for i in range(n):
        sd = O.Seed((i * 2246822519 + 99) & 0xFFFFFFFF, (i * 3266489917 + 7) & 0xFFFFFFFF)
        O.lib.lpto_sample_ggx_vndf(O.v3(wi), alpha, C.byref(sd), out)
        ms[i] = out[0], out[1], out[2]
    assert np.allclose(np.linalg.norm(ms, axis=1), 1.0, atol=1e-5) and np.all(ms[:, 2] >= 0)
    assert np.all(ms @ wi > -1e-6)  # only normals that face the viewer
    nz, nphi, sub = 6, 8, 128
    zi = np.minimum((ms[:, 2] * nz).astype(int), nz - 1)
    ph = np.mod(np.arctan2(ms[:, 1], ms[:, 0]), 2 * np.pi)
    pi_ = np.minimum((ph / (2 * np.pi) * nphi).astype(int), nphi - 1)
    counts = np.zeros((nz, nphi))
    np.add.at(counts, (zi, pi_), 1.0)
    z = (np.arange(nz * sub) + 0.5) / (nz * sub)
    phi = (np.arange(nphi * sub) + 0.5) / (nphi * sub) * 2 * np.pi
    Z, P = np.meshgrid(z, phi, indexing="ij")
    S = np.sqrt(1.0 - Z * Z)
    m = np.stack([S * np.cos(P), S * np.sin(P), Z], axis=-1)
    D = alpha * alpha / (np.pi * ((alpha * alpha - 1.0) * Z * Z + 1.0) ** 2)
    dv = np.maximum(m @ wi, 0.0) * D / (wi[2] * (1.0 + _ggx_lambda(wi, alpha)))
    cell = (1.0 / (nz * sub)) * (2 * np.pi / (nphi * sub))
    model = (dv * cell).reshape(nz, sub, nphi, sub).sum(axis=(1, 3))
    assert abs(model.sum() - 1.0) < 0.02  # the quadrature resolves the lobe
    sigma = np.sqrt(model * (1.0 - model) / n)
    err = np.abs(counts / n - model)
    assert np.all(err <= 4.5 * sigma + 0.004), float((err - 4.5 * sigma).max())


def test_free_paths_in_the_microsurface_are_exponential_in_height():
    """Smith's model with an exponential height distribution: going up along w the chance to travel a height
    x without meeting the surface is exp(-Lambda(w) x); going down the rate is 1 + Lambda(-w)."""
    alpha = 0.6
    n = 40000
    for w in ((0.6, 0.0, 0.8), (0.95, 0.1, 0.29580399), (0.0, 0.0, 1.0), (0.6, 0.0, -0.8), (0.1, 0.9, -0.42426407)):
        w = np.array(w) / np.linalg.norm(w)
        up = w[2] > 0
        lam = float(_ggx_lambda(np.array([w[0], w[1], abs(w[2])]), alpha))
        rate = lam if up else 1.0 + lam
        d = np.empty(n)
        for i in range(n):
            sd = O.Seed((i * 2654435761 + 5) & 0xFFFFFFFF, (i * 97 + 31) & 0xFFFFFFFF)
            d[i] = O.lib.lpto_sample_ggx_height(O.v3(w), -2.0, alpha, C.byref(sd)) + 2.0
        assert np.all(d >= 0) if up else np.all(d <= 0)
        x = np.abs(d)
        if rate < 1e-6:  # straight up: Lambda = 0 (the shader floors the projected area at 1e-7, heitz/BSDF.glsl:74): the ray leaves at once
            assert np.median(x) > 1e6
            continue
        assert abs(x.mean() * rate - 1.0) < 4.5 / np.sqrt(n)
        for q in (0.5, 2.0):
            want = np.exp(-q)
            assert abs((x * rate > q).mean() - want) < 4.5 * np.sqrt(want * (1 - want) / n)


def test_smooth_diffuse_is_lambertian():
    for view in ((0.0, 0.0, 1.0), (0.8, 0.0, 0.6)):
        dirs, wts = _sample_bsdf(O.K_DIFF, view, 0.01, 1.5, 20000, base=(0.7, 0.7, 0.7))
        z = dirs[:, 2]
        assert np.all(z > 0)
        # cosine-weighted hemisphere whatever the view: E[cos] = 2/3, E[cos^2] = 1/2, azimuth uniform
        assert abs(z.mean() - 2.0 / 3.0) < 0.006 and abs((z * z).mean() - 0.5) < 0.008
        assert abs(dirs[:, 0].mean()) < 0.012 and abs(dirs[:, 1].mean()) < 0.012
        assert np.allclose(wts[wts[:, 0] > 0], 0.7, atol=1e-6)  # one scattering event: albedo once


# ---- the estimator ----------------------------------------------------------------------------------------

def test_independent_renders_agree_and_noise_falls_as_one_over_sqrt_n(cornell, cornell_oracle):
    w, h, bounces = 48, 48, 6

    def mean_image(seed, spp):
        img, _ = O.render_frames(cornell_oracle, L.frame_ubos(cornell.camera_world, cornell.fov_y, spp, seed=seed), w, h, bounces)
        assert np.all(img[..., 3] == spp)
        return img[..., :3].astype(np.float64) / spp

    a64, b64 = mean_image(101, 64), mean_image(202, 64)
    a16, b16 = mean_image(303, 16), mean_image(404, 16)
    # the image means of independent runs agree within a few percent (same expectation)
    assert abs(a64.mean() / b64.mean() - 1) < 0.03
    # the difference of two independent N-sample means has variance 2 sigma^2 / N
    v64, v16 = ((a64 - b64) ** 2).mean(), ((a16 - b16) ** 2).mean()
    assert 2.5 < v16 / v64 < 6.0, v16 / v64
    # 64 frames in one call = 4 calls of 16 frames accumulated (float addition order is the frame order)
    ubos = L.frame_ubos(cornell.camera_world, cornell.fov_y, 64, seed=101)
    once, _ = O.render_frames(cornell_oracle, ubos, w, h, bounces)
    acc = None
    for k in range(4):
        acc, _ = O.render_frames(cornell_oracle, ubos[16 * k:16 * k + 16], w, h, bounces, accum=acc)
    assert np.array_equal(once, acc)


def test_deeper_paths_only_add_light_in_the_closed_part_of_the_box(cornell, cornell_oracle):
    """More bounces gather more of the emitter's light (energy is non-negative and the series converges):
    the mean image grows from 2 to 4 to 8 bounces and the increments shrink."""
    w = h = 40
    ubos = L.frame_ubos(cornell.camera_world, cornell.fov_y, 48, seed=7)
    # pixels that see the inside of the box (the open front shows the 0.2 environment)
    means = []
    for bounces in (2, 4, 8, 16):
        img, _ = O.render_frames(cornell_oracle, ubos, w, h, bounces)
        means.append(img[..., :3].astype(np.float64).mean() / 48)
    gains = np.diff(means)
    assert np.all(gains > -0.01 * means[-1])
    assert gains[0] > gains[2] - 0.01 * means[-1]


def test_libm_variant_converges_to_the_same_image(cornell, cornell_oracle):
    """GLSL leaves the precision of sin, cos, tan, log, exp and pow open (the calls: heitz/BSDF.glsl:55-56,81,
    common/util.glsl:7, path_tracing.comp:119); the arithmetic contract fixes them as polynomials.  A second build
    of the restatement calls the C library instead (oracle/Makefile, -DLPTO_LIBM) — another legitimate
    implementation of the same shader.  Rendered with the same seeds, the two differ only where a rounding
    difference flips a branch; such flips must not bias the estimator: the converged images agree within Monte-Carlo
    noise, pixel by pixel, and far more closely than two independent renders do."""
    w = h = 128
    bounces, chunks, per_chunk = 8, 16, 256  # 4096 samples per pixel
    libm = O.libm_variant()
    # the two builds do differ: the C library's functions are not the polynomials
    x = np.float32(0.7)
    assert libm.lpto_log(x) != 0.0 and abs(libm.lpto_log(x) - O.lib.lpto_log(x)) < 1e-6
    ubos = L.frame_ubos(cornell.camera_world, cornell.fov_y, chunks * per_chunk, seed=0x5EED)
    a = np.empty((chunks, h, w, 3))
    b = np.empty((chunks, h, w, 3))
    seg_a = seg_b = 0
    for k in range(chunks):
        u = ubos[k * per_chunk:(k + 1) * per_chunk].copy()
        u["reset"] = 0
        u["reset"][0] = 1
        ia, sa = O.render_frames(cornell_oracle, u, w, h, bounces)
        ib, sb = O.render_frames(cornell_oracle, u, w, h, bounces, library=libm)
        assert np.all(ia[..., 3] == per_chunk) and np.all(ib[..., 3] == per_chunk)
        a[k] = ia[..., :3].astype(np.float64) / per_chunk
        b[k] = ib[..., :3].astype(np.float64) / per_chunk
        seg_a += sa["segments"]
        seg_b += sb["segments"]
    # same seeds: nearly all paths are the same paths (the segment counts differ by a tiny fraction) ...
    assert seg_a != seg_b or not np.array_equal(a, b)  # ... but not all: some branches did flip
    assert abs(seg_a - seg_b) < 1e-4 * seg_a  # measured: 2e-6
    mean_a, mean_b = a.mean(axis=0), b.mean(axis=0)
    # image means: equal to a small fraction of a percent
    assert abs(mean_a.mean() / mean_b.mean() - 1.0) < 5e-4  # measured: 7e-6
    # paired test per pixel and channel over the chunks: d_k = a_k - b_k has mean zero under the hypothesis
    d = a - b
    sd = d.std(axis=0, ddof=1)
    live = sd > 0  # pixels where the two builds gave different samples at all
    z = d.mean(axis=0)[live] / (sd[live] / np.sqrt(chunks))
    assert live.mean() > 0.1  # the test has something to look at (a pixel with a single flip gives z = +-1)
    assert abs(z.mean()) < 0.1, z.mean()  # no systematic sign
    assert 0.7 < z.std() < 1.5, z.std()  # Student-t with 15 degrees of freedom: 1.07
    assert (np.abs(z) > 5.0).mean() < 2e-3
    # and against the noise of the render itself: the difference of the two builds is far smaller than the difference
    # of two independent halves of one build
    half = chunks // 2
    independent = ((a[:half].mean(axis=0) - a[half:].mean(axis=0)) ** 2).mean()
    paired = ((mean_a - mean_b) ** 2).mean()
    assert paired < 0.02 * independent, (paired, independent)  # measured: 3e-4
    # relative L2 of the converged images
    assert np.sqrt(paired * mean_a.size) / np.sqrt((mean_a ** 2).sum()) < 0.005

"""CPU suite: the oracle (oracle/) against fixtures captured from the reference.

Gate: rel-L2 <= 1e-5 for fp32 image work (the reference's own fp32 reduction
order is the noise floor), exact for tables, masks and indices.
"""
import numpy as np
import pytest

from standin import rel_l2

TOL = 1e-5


# ----------------------------------------------------------------- tables
@pytest.mark.parametrize("tag,resp", [("full", ""), ("r20", "20"), ("r100", "100"), ("ddim50", "ddim50")])
def test_schedule_tables_exact(oracle, golden, tag, resp):
    g = golden("tables")
    s = oracle.tables.schedule(1000, "linear", resp)
    for k in ("betas", "alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
              "posterior_mean_coef1", "posterior_mean_coef2", "posterior_log_variance_clipped", "log_betas",
              "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "posterior_variance"):
        np.testing.assert_array_equal(s[k], g[f"{tag}.{k}"], err_msg=k)
    np.testing.assert_array_equal(s["timestep_map"], g[f"{tag}.timestep_map"])


def test_gaussian_kernel(oracle, golden):
    g = golden("tables")
    k = oracle.tables.gaussian_kernel2d(61, 3.0)
    np.testing.assert_allclose(k, g["gauss61_s3.kernel_f64"], rtol=0, atol=1e-17)
    np.testing.assert_array_equal(k.astype(np.float32), g["gauss61_s3.weight_f32"][0])
    assert (k > 0).sum() == 625


@pytest.mark.parametrize("f,hw", [(4, 256), (8, 256), (4, 64)])
def test_resize_tables_exact(oracle, golden, f, hw):
    g = golden("tables")
    t = oracle.tables.resize_tables(hw, hw, f)
    assert t["order"] == list(g[f"sr{f}_{hw}.sorted_dims"])
    for d, key in ((2, "h"), (3, "w")):
        np.testing.assert_array_equal(t["w_" + key], g[f"sr{f}_{hw}.w_dim{d}"])
        np.testing.assert_array_equal(t["i_" + key], g[f"sr{f}_{hw}.i_dim{d}"])


# ----------------------------------------------------------------- posterior step
@pytest.mark.parametrize("t", [999, 500, 1, 0])
def test_posterior_step(oracle, golden, t):
    g = golden("posterior")
    c = oracle.tables.step_coefs(oracle.tables.schedule(1000), t)
    f = oracle.posterior_fwd(g["x"], g[f"t{t}.model_out"], g[f"t{t}.noise"], c)
    # element-wise chain in the reference's op order: bit-exact except through exp()
    np.testing.assert_array_equal(f["x0_hat"], g[f"t{t}.x0_hat"])
    assert rel_l2(f["sample"], g[f"t{t}.sample"]) < 1e-6
    gx, gmo = oracle.posterior_bwd(g["w_x0"], g["w_s"], g["x"], g[f"t{t}.model_out"], g[f"t{t}.noise"], c)
    assert rel_l2(gx, g[f"t{t}.g_x"]) < TOL
    assert rel_l2(gmo, g[f"t{t}.g_model_out"]) < TOL
    clamped = np.abs(g[f"t{t}.x0_hat"]) == 1.0
    assert 0.02 < clamped.mean() < 0.9


# ----------------------------------------------------------------- operators
def _inputs(g, tag):
    if f"{tag}.x" in g.keys():
        return g[f"{tag}.x"], g[f"{tag}.u"]
    import torch
    gen = torch.Generator().manual_seed(int(g[f"{tag}.seed"]))
    shape = [int(v) for v in g[f"{tag}.shape"]]
    x = torch.rand(*shape, generator=gen) * 2 - 1
    u = torch.randn(*g[f"{tag}.y"].shape, generator=gen)
    return x.numpy(), u.numpy()


def _op(oracle, g, tag):
    base = tag.split(".")[0]
    if base == "gauss":
        return oracle.make_operator("gaussian_blur", kernel_size=61, intensity=3.0)
    if base == "motion":
        return oracle.make_operator("motion_blur", kernel=g["motion.kernel"])
    if base in ("sr4", "sr8"):
        hw = int(g[f"{tag}.shape"][-1])
        return oracle.make_operator("super_resolution", in_shape=(1, 3, hw, hw), scale_factor=int(base[2:]))
    if base == "inpaint":
        return oracle.make_operator("inpainting", mask=g["inpaint.mask"])
    if base == "phase":
        return oracle.make_operator("phase_retrieval", oversample=2.0)
    raise KeyError(tag)


@pytest.mark.parametrize("tag", ["gauss.small", "gauss.full", "motion.small", "sr4.small", "sr4.full",
                                 "sr8.full", "inpaint.small", "phase.small", "phase.full"])
def test_operator_forward_adjoint(oracle, golden, tag):
    g = golden("operators")
    x, u = _inputs(g, tag)
    op = _op(oracle, g, tag)
    y = op.forward(x)
    adj = op.adjoint(u, x.shape[-2:])
    if tag.startswith("inpaint"):
        np.testing.assert_array_equal(y, g[f"{tag}.y"])
        np.testing.assert_array_equal(adj, g[f"{tag}.adj"])
    else:
        assert rel_l2(y, g[f"{tag}.y"]) < TOL, "forward"
        assert rel_l2(adj, g[f"{tag}.adj"]) < TOL, "adjoint"


@pytest.mark.parametrize("name", ["gaussian_blur", "motion_blur", "super_resolution"])
def test_adjoint_identity(oracle, name):
    """<A x, u> == <x, A^T u> -- the size-independent property used at full size on the GPU."""
    from standin import synthetic_motion_kernel
    rng = np.random.RandomState(3)
    cfg = {"gaussian_blur": dict(kernel_size=61, intensity=3.0),
           "motion_blur": dict(kernel=synthetic_motion_kernel(61, 5)),
           "super_resolution": dict(in_shape=(1, 3, 48, 48), scale_factor=4)}[name]
    op = oracle.make_operator(name, **cfg)
    x = rng.randn(2, 3, 48, 48).astype(np.float32)
    y = op.forward(x)
    u = rng.randn(*y.shape).astype(np.float32)
    lhs = np.vdot(y.astype(np.float64), u.astype(np.float64))
    rhs = np.vdot(x.astype(np.float64), op.adjoint(u, (48, 48)).astype(np.float64))
    assert abs(lhs - rhs) <= 1e-6 * max(abs(lhs), 1.0)


def test_symmetric_kernel_adjoint_is_forward_on_a_doubled_border(oracle):
    """The identity k_blur_sep_adj_sym rests on (csrc/blur_sep.h): for symmetric taps, (C R)^T u = D C E u -- E the forward
    operator's reflection extension with the image's own border sample doubled, D halving the outputs on the border --
    against the dense matrix in one dimension, and against the oracle's adjoint of the BASELINE Gaussian (whose taps are
    bitwise symmetric) in two."""
    rng = np.random.default_rng(0)
    for n, R in ((16, 3), (9, 4), (64, 12), (7, 5), (14, 12)):
        half = rng.standard_normal(R + 1)
        taps = np.concatenate([half[:0:-1], half])
        refl = lambda p: -p if p < 0 else (2 * (n - 1) - p if p > n - 1 else p)
        A = np.zeros((n, n))
        for i in range(n):
            for d in range(2 * R + 1):
                A[i, refl(i + d - R)] += taps[d]
        u = rng.standard_normal(n)
        e = np.array([u[refl(q - R)] for q in range(n + 2 * R)])
        e[R] *= 2
        e[R + n - 1] *= 2
        out = np.array([np.dot(taps, e[m:m + 2 * R + 1]) for m in range(n)])
        out[0] *= 0.5
        out[n - 1] *= 0.5
        np.testing.assert_allclose(out, A.T @ u, rtol=0, atol=1e-12)
    k2 = oracle.tables.gaussian_kernel2d(61, 3.0).astype(np.float32)
    assert np.array_equal(k2, k2[::-1, :]) and np.array_equal(k2, k2[:, ::-1]) and np.array_equal(k2, k2.T)
    op = oracle.make_operator("gaussian_blur", kernel_size=61, intensity=3.0)
    u = rng.standard_normal((1, 2, 40, 56)).astype(np.float32)
    e = np.pad(u, ((0, 0), (0, 0), (30, 30), (30, 30)), mode="reflect").astype(np.float64)
    e[:, :, 30, :] *= 2; e[:, :, 30 + 39, :] *= 2; e[:, :, :, 30] *= 2; e[:, :, :, 30 + 55] *= 2
    out = np.zeros((1, 2, 40, 56))
    for dy in range(61):
        for dx in range(61):
            if k2[dy, dx] != 0:
                out += float(k2[dy, dx]) * e[:, :, dy:dy + 40, dx:dx + 56]
    out[:, :, 0, :] *= 0.5; out[:, :, -1, :] *= 0.5; out[:, :, :, 0] *= 0.5; out[:, :, :, -1] *= 0.5
    assert rel_l2(out, op.adjoint(u, (40, 56))) < 1e-6


# ----------------------------------------------------------------- conditioning per call
def _cond_op(oracle, g, oname):
    if oname == "gauss":
        return oracle.make_operator("gaussian_blur", kernel_size=61, intensity=3.0)
    if oname == "motion":
        return oracle.make_operator("motion_blur", kernel=golden_motion_kernel)
    if oname == "sr4":
        return oracle.make_operator("super_resolution", in_shape=(1, 3, 32, 32), scale_factor=4)
    if oname == "inpaint":
        return oracle.make_operator("inpainting", mask=g["inpaint.mask"])
    return oracle.make_operator("phase_retrieval", oversample=2.0)


golden_motion_kernel = None


@pytest.mark.parametrize("oname", ["gauss", "motion", "sr4", "inpaint", "phase"])
@pytest.mark.parametrize("t", [900, 500, 0])
def test_conditioning_per_call(oracle, golden, oname, t):
    """ps / ps_semantic(sem=0) / ps_anneal outputs of the reference for one step, teacher-forced:
    the UNet is the stand-in, whose VJP the test takes with torch (as the product does)."""
    import torch
    from standin import StandInModel
    global golden_motion_kernel
    golden_motion_kernel = golden("operators")["motion.kernel"]
    g = golden("conditioning")
    sched = oracle.tables.schedule(1000)
    c = oracle.tables.step_coefs(sched, t)
    op = _cond_op(oracle, g, oname)
    model = StandInModel()
    x_prev = torch.from_numpy(g[f"{oname}.t{t}.x_prev"]).requires_grad_()
    mo = model(x_prev, torch.tensor([float(t)]))

    def unet_vjp(g_mo):
        (gx,) = torch.autograd.grad(mo, x_prev, torch.from_numpy(g_mo), retain_graph=True)
        return gx.numpy()

    y, noise = g[f"{oname}.y"], g[f"{oname}.t{t}.noise"]
    xp, mo_np = x_prev.detach().numpy(), mo.detach().numpy()

    # ps_semantic(scale=.7, sem=0): returns the gradient itself
    s = oracle.dps_step(op, xp, mo_np, noise, y, c, scale=0.7, power=1, g_unet_fn=unet_vjp)
    np.testing.assert_array_equal(s["x0_hat"], g[f"{oname}.t{t}.x0_hat"])
    assert rel_l2(s["sample"], g[f"{oname}.t{t}.sample"]) < 1e-6
    assert rel_l2(s["norm"], g[f"{oname}.t{t}.ps_semantic.ret1"]) < TOL
    assert rel_l2(s["grad"], g[f"{oname}.t{t}.ps_semantic.ret0"]) < 2e-5
    s2 = oracle.dps_step(op, xp, mo_np, noise, y, c, scale=0.7, power=2, g_unet_fn=unet_vjp)
    # NB the reference's ps_semantic applies norm_exp to the *semantic* term only
    # (condition_methods.py:168-184): the measurement term stays first power.
    assert rel_l2(s["grad"], g[f"{oname}.t{t}.ps_semantic2.ret0"]) < 2e-5
    del s2

    # ps(scale=.3): x_t - scale*grad, norm, scale/2/norm
    s = oracle.dps_step(op, xp, mo_np, noise, y, c, scale=0.3, power=1, g_unet_fn=unet_vjp)
    assert rel_l2(s["x_next"], g[f"{oname}.t{t}.ps.ret0"]) < TOL
    assert rel_l2(s["norm"], g[f"{oname}.t{t}.ps.ret1"]) < TOL
    assert rel_l2(0.3 / 2 / s["norm"], g[f"{oname}.t{t}.ps.ret2"]) < TOL

    # ps_anneal: norm^2 gradient scaled by beta_t / (anneal * max(sigma, .05)^2)
    net = float(sched["betas"][t]) / (1.0 * 0.05 ** 2)
    s = oracle.dps_step(op, xp, mo_np, noise, y, c, scale=net, power=2, g_unet_fn=unet_vjp)
    assert rel_l2(s["x_next"], g[f"{oname}.t{t}.ps_anneal.ret0"]) < TOL
    assert rel_l2(net, g[f"{oname}.t{t}.ps_anneal.ret2"]) < 1e-6


# ----------------------------------------------------------------- select
def test_argmin_gather(oracle, golden):
    v = np.array([3.0, 1.0, 1.0, 7.0], dtype=np.float32)
    assert oracle.argmin(v) == 1
    v[2] = np.nan
    assert oracle.argmin(v) == 2
    src = np.arange(4 * 6, dtype=np.float32).reshape(4, 1, 2, 3)
    out = oracle.gather(src, [2, 2, 0])
    np.testing.assert_array_equal(out, src[[2, 2, 0]])


# ----------------------------------------------------------------- DDIM step (scope row f2)
DDIM_CASES = [(999, 0.0), (500, 0.0), (1, 0.0), (0, 0.0), (500, 0.5), (0, 0.5)]


@pytest.mark.parametrize("t,eta", DDIM_CASES)
def test_ddim_step_golden(oracle, golden, t, eta):
    """oracle DDIM variant of S1 (gaussian_diffusion.py:479-509) against the reference's p_sample and its autograd"""
    g = golden("ddim")
    tag = f"t{t}.eta{eta:g}"
    c = oracle.tables.ddim_step_coefs(oracle.tables.schedule(1000), t, eta)
    o = oracle.posterior_fwd(g["x"], g[f"{tag}.model_out"], g[f"{tag}.noise"], c)
    np.testing.assert_array_equal(o["x0_hat"], g[f"{tag}.x0_hat"])
    assert rel_l2(o["sample"], g[f"{tag}.sample"]) < 1e-6
    gx, gmo = oracle.posterior_bwd(g["w_x0"], g["w_s"], g["x"], g[f"{tag}.model_out"], g[f"{tag}.noise"], c)
    assert rel_l2(gx, g[f"{tag}.g_x"]) < 1e-5
    assert rel_l2(gmo, g[f"{tag}.g_model_out"]) < 1e-5
    assert np.all(gmo[:, 3:] == 0)          # DDIM ignores the variance channels


@pytest.mark.parametrize("tag,opname,cfg", [("gauss", "gaussian_blur", dict(kernel_size=61, intensity=3.0)),
                                            ("sr4", "super_resolution", dict(in_shape=(1, 3, 64, 64), scale_factor=4))])
def test_resample_update_golden(oracle, golden, tag, opname, cfg):
    """SearchDDPM.resample_update of the reference (gaussian_diffusion.py:515-587), all potential types, first call /
    update / resample + update: the oracle's cost update reproduces the reference's net costs for the reference's draw"""
    g = golden("resample")
    op = oracle.make_operator(opname, **cfg)
    for pot in ("mean", "min", "diff", "curr"):
        for case in ("first", "noresample", "resample"):
            ids = g[f"{tag}.{pot}.{case}.ids"]
            prev = None if case == "first" else g[f"{tag}.prev_costs"]
            cands, net = oracle.resample_update(op, g[f"{tag}.candidates"], g[f"{tag}.denoised"], g[f"{tag}.y"], prev,
                                                pot, ids=ids if case == "resample" else None)
            assert rel_l2(net, g[f"{tag}.{pot}.{case}.net"]) < TOL, (pot, case)
            np.testing.assert_array_equal(cands[:, 0, 0, 0].round().astype(np.int64), ids)
    assert list(g["gauss.flat.ids"]) == list(range(6))


@pytest.mark.parametrize("name", ["gaussian_blur", "super_resolution", "inpainting", "phase_retrieval"])
def test_torch_ops_reference_agrees_with_the_port(oracle, name):
    """oracle/torch_ref.py (the reference's ATen ops, bench.py's second CPU baseline) against the C port on one `ps` step"""
    from oracle import torch_ref
    rng = np.random.RandomState(3)
    n, hw = 2, 64
    c = oracle.tables.step_coefs(oracle.tables.schedule(1000), 400)
    x = rng.randn(n, 3, hw, hw).astype(np.float32)
    eps = ((c["a"] * x - 1.3 * np.tanh(rng.randn(n, 3, hw, hw))) / c["b"]).astype(np.float32)
    mo = np.concatenate([eps, rng.uniform(-1, 1, eps.shape).astype(np.float32)], axis=1)
    z = rng.randn(n, 3, hw, hw).astype(np.float32)
    gu = (1e-2 * rng.randn(n, 3, hw, hw)).astype(np.float32)
    mask = (rng.rand(1, 1, hw, hw) < 0.5).astype(np.float32)
    cfg = {"gaussian_blur": dict(kernel_size=61, intensity=3.0), "super_resolution": dict(in_shape=(1, 3, hw, hw), scale_factor=4),
           "inpainting": dict(mask=mask), "phase_retrieval": dict(oversample=2.0)}[name]
    orc = oracle.make_operator(name, **cfg)
    tkw = {"gaussian_blur": dict(kernel=orc.kw.get("kernel")), "super_resolution": dict(tables=orc.kw.get("tables")),
           "inpainting": dict(mask=mask), "phase_retrieval": dict(pad=64)}[name]
    y = orc.forward(rng.uniform(-1, 1, (1, 3, hw, hw)).astype(np.float32))
    y = (y + 0.05 * rng.randn(*y.shape)).astype(np.float32)
    ref = oracle.dps_step(orc, x, mo, z, y, c, scale=0.3, power=1, g_unet_fn=lambda g: gu)
    got = torch_ref.dps_step(torch_ref.TorchOperator(name, **tkw), x, mo, z, y, c, 0.3, gu)
    assert np.array_equal(got["x0_hat"], ref["x0_hat"])
    for k in ("sample", "norm", "x_next"):
        assert rel_l2(got[k], ref[k]) < 1e-5, k

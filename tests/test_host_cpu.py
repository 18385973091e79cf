"""CPU suite: host-side logic of the product (tables, registries, error behaviour, fusion planning)."""
import functools

import numpy as np
import pytest
import torch

DIFF = dict(steps=1000, noise_schedule="linear", model_mean_type="epsilon", model_var_type="learned_range",
            dynamic_threshold=False, clip_denoised=True, rescale_timesteps=True)


@pytest.mark.parametrize("tag,resp", [("full", ""), ("r20", "20"), ("r100", "100"), ("ddim50", "ddim50")])
def test_sampler_tables_match_reference_exactly(golden, tag, resp):
    from dps_ttc_amd.gaussian_diffusion import create_sampler
    g = golden("tables")
    s = create_sampler(sampler="ddpm", timestep_respacing=resp, **DIFF)
    for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
              "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
              "posterior_mean_coef1", "posterior_mean_coef2", "posterior_variance",
              "posterior_log_variance_clipped"):
        np.testing.assert_array_equal(getattr(s, k), g[f"{tag}.{k}"], err_msg=k)
    assert list(s.timestep_map) == list(g[f"{tag}.timestep_map"])
    assert s.num_timesteps == len(g[f"{tag}.betas"])
    c = s.step_coefs[s.num_timesteps - 1]
    assert c.a == np.float32(g[f"{tag}.sqrt_recip_alphas_cumprod"][-1]) and c.add_noise == 1
    assert s.step_coefs[0].add_noise == 0
    assert s.step_coefs[3].max_log == np.float32(g[f"{tag}.log_betas"][3])


def test_host_tables_match_reference_exactly(golden):
    from dps_ttc_amd import host_tables
    g = golden("tables")
    k = host_tables.gaussian_blur_kernel(61, 3.0)
    np.testing.assert_allclose(k, g["gauss61_s3.kernel_f64"], rtol=0, atol=2e-18)
    np.testing.assert_array_equal(k.astype(np.float32), g["gauss61_s3.weight_f32"][0])
    for f, hw in ((4, 256), (8, 256), (4, 64)):
        w, i = host_tables.resizer_axis(hw, 1.0 / f)
        np.testing.assert_array_equal(w, g[f"sr{f}_{hw}.w_dim2"])
        np.testing.assert_array_equal(i, g[f"sr{f}_{hw}.i_dim3"])


def test_registries_and_errors():
    from dps_ttc_amd import condition_methods as CM
    from dps_ttc_amd import gaussian_diffusion as GD
    from dps_ttc_amd import measurements as MS
    assert sorted(MS.__OPERATOR__) == ["gaussian_blur", "inpainting", "motion_blur", "noise", "nonlinear_blur",
                                       "phase_retrieval", "super_resolution"]
    assert sorted(MS.__NOISE__) == ["clean", "gaussian", "poisson"]
    assert sorted(CM.__CONDITIONING_METHOD__) == ["mcg", "projection", "ps", "ps+", "ps_anneal", "ps_semantic", "vanilla"]
    assert sorted(GD.__SAMPLER__) == ["ddim", "ddpm", "search_ddpm", "ttc_ddim"]
    with pytest.raises(NameError):
        MS.get_operator("no_such_operator", device="cpu")
    with pytest.raises(NameError):
        MS.get_noise("no_such_noise")
    with pytest.raises(NameError):
        CM.get_conditioning_method("no_such_method", None, None)
    with pytest.raises(NameError):
        GD.create_sampler(sampler="no_such_sampler", **DIFF)
    with pytest.raises(NameError):
        MS.register_operator("gaussian_blur")(object)            # duplicate registration
    with pytest.raises(NotImplementedError):
        MS.get_operator("nonlinear_blur", opt_yml_path="x.yml", device="cpu")
    noiser = MS.get_noise("gaussian", sigma=0.05)
    assert noiser.__name__ == "gaussian" and noiser.sigma == 0.05
    assert MS.get_noise("clean")(torch.ones(2)).equal(torch.ones(2))
    torch.manual_seed(0)
    y = noiser(torch.zeros(1000))
    assert abs(float(y.std()) - 0.05) < 0.01


def test_operators_have_no_cpu_fallback():
    from dps_ttc_amd.measurements import get_operator
    op = get_operator("gaussian_blur", kernel_size=61, intensity=3.0, device="cpu")
    assert tuple(op.get_kernel().shape) == (1, 1, 61, 61) and op.get_kernel().dtype == torch.float64
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        op.forward(torch.zeros(1, 3, 64, 64))
    sr = get_operator("super_resolution", in_shape=(1, 3, 64, 64), scale_factor=4, device="cpu")
    assert sr.w_w.shape == (16, 16) and sr.i_h.dtype == np.int64
    assert tuple(sr.transpose(torch.zeros(1, 3, 16, 16)).shape) == (1, 3, 64, 64)      # nearest upsample, as the reference
    inp = get_operator("inpainting", device="cpu")
    with pytest.raises(ValueError, match="Require mask"):
        inp.forward(torch.zeros(1, 3, 8, 8))
    pr = get_operator("phase_retrieval", oversample=2.0, device="cpu")
    assert pr.pad == 64


def test_motion_blur_kernel_plumbing():
    from dps_ttc_amd.measurements import get_operator
    np.random.seed(3)
    a = get_operator("motion_blur", kernel_size=61, intensity=0.5, device="cpu")
    np.random.seed(3)
    b = get_operator("motion_blur", kernel_size=61, intensity=0.5, device="cpu")
    np.testing.assert_array_equal(a.kernel_matrix, b.kernel_matrix)          # np.random.seed(kernel_idx) selects it
    assert abs(a.kernel_matrix.sum() - 1.0) < 1e-12 and tuple(a.get_kernel().shape) == (1, 1, 61, 61)
    k = np.arange(9, dtype=np.float32).reshape(3, 3)
    a.set_kernel(k)
    np.testing.assert_array_equal(a._weights, k.T)                             # stored transposed (measurements.py:125)


def test_fused_spec_and_fusion_planning():
    from dps_ttc_amd.condition_methods import get_conditioning_method
    from dps_ttc_amd.gaussian_diffusion import GaussianDiffusion, create_sampler
    from dps_ttc_amd.measurements import get_noise, get_operator
    op = get_operator("gaussian_blur", kernel_size=61, intensity=3.0, device="cpu")
    g = get_noise("gaussian", sigma=0.01)
    ps = get_conditioning_method("ps", op, g, scale=0.3)
    assert ps.fused_spec() == {"scale": 0.3, "power": 1} and not ps.returns_gradient
    sem = get_conditioning_method("ps_semantic", op, g, scale=0.7, sem_guid_scale=0.0, norm_exp=2)
    assert sem.fused_spec() == {"scale": 0.7, "power": 1} and sem.returns_gradient      # measurement term stays first power
    ann = get_conditioning_method("ps_anneal", op, g, scale=0.3)
    assert ann.noise_sigma == 0.05
    assert ann.fused_spec(beta_scale=0.02) == {"scale": 0.02 / 0.05 ** 2, "power": 2}
    assert get_conditioning_method("ps", op, get_noise("poisson", rate=1.0)).fused_spec() is None
    assert get_conditioning_method("mcg", op, g).fused_spec() is None
    assert abs(sem.semantic_scale(1.0) - 0.0) < 1e-12
    sem2 = get_conditioning_method("ps_semantic", op, g, sem_guid_scale=0.01, anneal_factor=10.0, embedder=lambda x: x)
    assert abs(sem2.semantic_scale(0.0) - 0.01 * (1 + 9 / (1 + np.exp(-3.0)))) < 1e-12
    spec2 = sem2.fused_spec(t=0.5)            # active semantic term: its x0_hat cotangent rides the fused backward
    assert spec2["scale"] == sem2.scale and spec2["power"] == 1 and callable(spec2["semantic"])
    sem2.guid_image_emb = torch.zeros(1, 1, 3 * 4 * 4)
    sem2.embedder = lambda x: x.reshape(x.shape[0], -1)
    x0 = torch.randn(2, 3, 4, 4, generator=torch.Generator().manual_seed(0))
    g, d = spec2["semantic"](x0)
    # loss = s_t * ||x0||_2 per particle  ->  gradient s_t * x0 / ||x0||
    st = sem2.semantic_scale(0.5)
    ref = st * x0 / x0.reshape(2, -1).norm(dim=1).view(2, 1, 1, 1)
    assert torch.allclose(g, ref, atol=1e-6) and torch.allclose(d, x0.reshape(2, -1).norm(dim=1), atol=1e-6)
    m, kw = GaussianDiffusion._unwrap_cond_fn(functools.partial(functools.partial(ps.conditioning, mask=1), l1=2))
    assert m is ps and kw == {"mask": 1, "l1": 2}
    m, kw = GaussianDiffusion._unwrap_cond_fn(lambda **k: None)
    assert m is None
    ddim = create_sampler(sampler="ddim", **DIFF, timestep_respacing="")
    mcg = get_conditioning_method("mcg", op, g)
    assert ddim._fusion_plan(mcg.conditioning, torch.zeros(1, 3, 64, 64)) is None       # two-value methods: per-op path
    # the DDIM record of a step: same a, b; c1 = sqrt(abar_prev), c2 = sqrt(1 - abar_prev - sigma^2), min_log = sigma
    import oracle.tables as T
    sched = T.schedule(1000)
    for t, eta in ((999, 0.0), (500, 0.5), (0, 0.0), (0, 0.5)):
        c, ref = ddim.sample_coefs(t, eta), T.ddim_step_coefs(sched, t, eta)
        for k in ("a", "b", "c1", "c2", "min_log"):
            assert np.float32(getattr(c, k)) == np.float32(ref[k]), (t, eta, k)
        assert c.add_noise == ref["add_noise"] == (2 | int(t != 0))
    other = create_sampler(sampler="ddpm", steps=1000, noise_schedule="linear", model_mean_type="start_x",
                           model_var_type="fixed_small", dynamic_threshold=False, clip_denoised=True,
                           rescale_timesteps=True, timestep_respacing="")
    assert not other.hip_posterior


def test_loops_refuse_cpu_inputs():
    from dps_ttc_amd.gaussian_diffusion import create_sampler
    s = create_sampler(sampler="ddpm", timestep_respacing="20", **DIFF)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        s.p_sample_loop(model=None, x_start=torch.zeros(1, 3, 8, 8), measurement=torch.zeros(1, 3, 8, 8),
                        measurement_cond_fn=None, record=False, save_root=None)


def test_space_timesteps_and_schedules():
    from dps_ttc_amd.gaussian_diffusion import get_named_beta_schedule, space_timesteps
    assert space_timesteps(300, [10, 15, 20]) == space_timesteps(300, "10,15,20")
    assert len(space_timesteps(1000, "ddim50")) == 50 and len(space_timesteps(1000, "250")) == 250
    with pytest.raises(ValueError):
        space_timesteps(10, [20])
    with pytest.raises(NotImplementedError):
        get_named_beta_schedule("no_such_schedule", 10)
    cos = get_named_beta_schedule("cosine", 100)
    assert cos.shape == (100,) and 0 < cos.min() and cos.max() <= 0.999


def test_diffstategrad_rank_rule_and_projection(golden):
    """reference diffstategrad_utils.py:4-78: the (flattened-cumsum) rank rule against values captured from the
    reference, and the projector's algebra (idempotent, identity off-period)"""
    from dps_ttc_amd.diffstategrad_utils import apply_diffstategrad, compute_svd_and_adaptive_rank
    g = golden("project")
    z = torch.from_numpy(g["rank.z"])
    for cutoff in (0.99, 0.9, 0.5):
        U, s, Vh, rank = compute_svd_and_adaptive_rank(z, cutoff)
        assert rank == int(g[f"rank.cut{cutoff:g}"])
    grad = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(1))
    assert apply_diffstategrad(grad, 7, 5, U, s, Vh, rank) is grad                 # off-period: untouched
    assert apply_diffstategrad(grad, 10, 0, U, s, Vh, rank) is grad                # period 0: never
    p1 = apply_diffstategrad(grad, 10, 5, U, s, Vh, rank)
    assert p1.shape == (1, 3, 64, 64)                                              # batch-0 only, as the reference
    p2 = apply_diffstategrad(p1, 10, 5, U, s, Vh, rank)
    assert float((p1 - p2).norm() / p1.norm()) < 1e-5
    with pytest.raises(ValueError):
        apply_diffstategrad(grad, 10, 5)

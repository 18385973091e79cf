#!/usr/bin/env python3
"""Driver counterpart of the reference's sample_condition_batched_ttc.py (same CLI, same three YAML files, same
output tree) on the MI355X hot path.

    python sample_condition_batched_ttc.py --model_config=configs/model_config.yaml \
        --diffusion_config=configs/diffusion_config.yaml --task_config=configs/gaussian_deblur_config.yaml \
        --n_paths=64 --batch_size=64 --ref_image_idxs=0 --gpu=0

For every reference image: y = A(x) + n once, then `n_paths // batch_size` particle groups of `batch_size`
particles through `sampler.p_sample_loop` (fused DPS loop for `sampler: ddpm`, per-step best-of-N for
`sampler: search_ddpm`), PSNR per particle, PNGs of input / label / every path, and the best-of-N pick by
measurement distance (best_of_n_simple.py semantics, on device).

Differences from the reference script, all fixes of things that crash there: `--l1` exists,
the sampler returns a tensor for this call signature, LPIPS is logged only if torchmetrics is installed.
With `torchrun --nproc-per-node G` (the reference shards by hand: run0.sh:12 / run1.sh:13 start one process per GPU
with its own --path_start_idx) the particle groups are sharded contiguously over the ranks and the best-of-N pick is
global: RCCL all-gather of the distances, winner broadcast from its owner.  `sampler: search_ddpm` then selects over
all ranks' particles at every step and `sampler: ttc_ddim` resamples over all of them (SURVEY.md 8e ii-iii).
"""
import argparse
import os
from functools import partial

# read by the HSA runtime when it initialises (first HIP call): must be in the environment before that, i.e. before anything
# below touches the GPU -- the host driver only supports dmabuf IPC, RCCL fails with hipIpcGetMemHandle otherwise
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")          # one hardware queue per HIP stream (particle groups, RCCL)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402

from dps_ttc_amd import distributed as dd  # noqa: E402
from dps_ttc_amd.condition_methods import get_conditioning_method  # noqa: E402
from dps_ttc_amd.data import get_dataloader, get_dataset, to_minus1_1  # noqa: E402
from dps_ttc_amd.gaussian_diffusion import create_sampler  # noqa: E402
from dps_ttc_amd.img_utils import clear_color, mask_generator  # noqa: E402
from dps_ttc_amd.measurements import get_noise, get_operator  # noqa: E402
from dps_ttc_amd.metrics import compute_psnr  # noqa: E402
from dps_ttc_amd.unet import create_model  # noqa: E402


def load_yaml(file_path: str) -> dict:
    with open(file_path) as f:
        return yaml.load(f, Loader=yaml.FullLoader)     # the task files carry !!python/tuple tags


def get_logger():
    import logging
    logger = logging.getLogger(name='DPS')
    if not logger.handlers:
        logger.setLevel(logging.INFO)
        h = logging.StreamHandler()
        h.setFormatter(logging.Formatter("%(asctime)s [%(name)s] >> %(message)s"))
        logger.addHandler(h)
    return logger


def imsave(path, array):
    try:
        import matplotlib.pyplot as plt
        plt.imsave(path, array)
    except ImportError:
        from PIL import Image
        a = (np.clip(array, 0, 1) * 255).astype(np.uint8)
        Image.fromarray(a).save(path)


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--model_config', type=str)
    p.add_argument('--diffusion_config', type=str)
    p.add_argument('--task_config', type=str)
    p.add_argument('--gpu', type=int, default=0)
    p.add_argument('--save_dir', type=str, default='./results_search')
    p.add_argument('--n_data_samples', type=int, default=1)
    p.add_argument('--n_paths', type=int, default=1)
    p.add_argument('--resample_every_steps', type=int, default=10)
    p.add_argument('--potential_type', type=str, default='curr')
    p.add_argument('--rs_temp', type=float, default=0.1)
    p.add_argument('--start_idx', type=int, default=0)
    p.add_argument('--path_start_idx', type=int, default=0)
    p.add_argument('--batch_size', type=int, default=1)
    p.add_argument('--anneal_scale', type=float, default=10)
    p.add_argument('--anneal_amp', type=float, default=1)
    p.add_argument('--anneal_loc', type=float, default=0.5)
    p.add_argument('--kernel_idx', type=int, default=0)
    p.add_argument('--ref_image_idxs', type=str, default='4')
    # additions
    p.add_argument('--l1', type=float, default=0.0, help='(the reference reads args.l1 without defining it)')
    p.add_argument('--timestep_respacing', type=str, default=None, help='override the diffusion YAML (e.g. "100")')
    p.add_argument('--seed', type=int, default=None)
    p.add_argument('--embedder', type=str, default=None,
                   help="'module:factory' -- factory(device) returns the embedding network of the semantic-guidance term "
                        "(ps_semantic with sem_guid_scale != 0); default: facenet_pytorch's InceptionResnetV1 as in the reference")
    p.add_argument('--guid_image', type=str, default=None,
                   help='guidance image for --embedder (PNG, 256x256); default: the reference image itself (oracle guidance, '
                        'for plumbing runs)')
    p.add_argument('--particle_groups', type=int, default=1,
                   help='run the batch_size particles of a fused DPS loop as this many independent sub-batches, each on its '
                        'own HIP stream with its own operator handle (kernels.ParticleGroups; results per particle unchanged)')
    return p.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    logger = get_logger()
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    if world > 1:
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", 0)) % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        backend = os.environ.get("DPSX_DIST_BACKEND", "nccl")      # "nccl" is RCCL; "gloo" only to rehearse on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        args.gpu = local
    if not torch.cuda.is_available():
        raise SystemExit("dps_ttc_amd runs the DPS hot path on an MI355X only (no CPU fallback by design)")
    device = torch.device(f"cuda:{args.gpu}")
    torch.cuda.set_device(device)
    logger.info(f"Device set to {device}.")

    model_config = load_yaml(args.model_config)
    diffusion_config = load_yaml(args.diffusion_config)
    task_config = load_yaml(args.task_config)
    if args.timestep_respacing is not None:
        diffusion_config['timestep_respacing'] = args.timestep_respacing
    if args.seed is not None:
        torch.manual_seed(args.seed + rank)

    model = create_model(**model_config).to(device).eval()

    measure_config = task_config['measurement']
    np.random.seed(args.kernel_idx)                     # selects the motion kernel / inpainting mask
    operator = get_operator(device=device, **measure_config['operator'])
    noiser = get_noise(**measure_config['noise'])
    op_name = measure_config['operator']['name']
    logger.info(f"Operation: {op_name} / Noise: {measure_config['noise']['name']}")

    cond_config = task_config['conditioning']
    cond_params = dict(cond_config['params'])
    embedder = None
    if args.embedder is not None:
        import importlib
        mod_name, _, fn_name = args.embedder.partition(':')
        embedder = getattr(importlib.import_module(mod_name), fn_name)(device)
        cond_params['embedder'] = embedder
    cond_method = get_conditioning_method(cond_config['method'], operator, noiser, **cond_params)
    measurement_cond_fn = cond_method.conditioning
    logger.info(f"Conditioning method : {cond_config['method']}")
    logger.info(f"Sampling: {diffusion_config['sampler']} / Steps: {diffusion_config['steps']}")

    sampler = create_sampler(**diffusion_config)
    sampler.particle_groups = max(1, args.particle_groups)
    groups = args.n_paths // args.batch_size
    if world > 1 and diffusion_config['sampler'] in ('search_ddpm', 'ttc_ddim'):
        # these loops exchange particles at every select / resample point: every rank runs the same number of groups
        if groups % world != 0:
            raise SystemExit(f"sampler {diffusion_config['sampler']} on {world} ranks needs n_paths / batch_size "
                             f"(= {groups} particle groups) to be a multiple of the number of ranks")
        if diffusion_config['sampler'] == 'search_ddpm':
            sampler.global_select = dd.GlobalSelect()
        else:
            sampler.global_resample = True
            # same stream on all ranks; a DEVICE generator: the draw runs on the GPU (as the reference's does), no host read
            sampler.resample_generator = torch.Generator(device=device).manual_seed(args.seed or 0)
    sample_fn = partial(sampler.p_sample_loop, model=model, measurement_cond_fn=measurement_cond_fn,
                        operator=operator, resample_every_steps=args.resample_every_steps,
                        potential_type=args.potential_type, rs_temp=args.rs_temp, anneal_scale=args.anneal_scale,
                        anneal_loc=args.anneal_loc, anneal_amp=args.anneal_amp)

    sigma = measure_config['noise'].get('sigma', 0)
    if cond_config['method'] == 'ps_anneal':
        dir_name = f"{op_name}_noise_sigma_{sigma}_dps_anneal_amp_{args.anneal_amp}"
    else:
        dir_name = f"{op_name}_noise_sigma_{sigma}_dps_scale_{cond_config['params']['scale']}"
    out_path = os.path.join(args.save_dir, dir_name)
    for img_dir in ['input', 'recon_paths', 'label', 'best_of_n']:
        os.makedirs(os.path.join(out_path, img_dir), exist_ok=True)

    data_config = task_config['data']
    dataset = get_dataset(**data_config, transforms=to_minus1_1)
    picks = [int(i) for i in args.ref_image_idxs.split(',')]
    subset = torch.utils.data.Subset(dataset, [min(i, len(dataset) - 1) for i in picks])
    loader = get_dataloader(subset, batch_size=1, num_workers=0, train=False)

    mask_gen = mask_generator(**measure_config['mask_opt']) if op_name == 'inpainting' else None
    if op_name == 'motion_blur' and rank == 0:
        imsave(os.path.join(out_path, f'kernel_{str(args.kernel_idx).zfill(5)}.png'), clear_color(operator.get_kernel()))

    # particle groups shard contiguously over the ranks: the rank-major order of the gathered scores is the path order
    g_lo, g_hi = dd.shard_range(groups, rank, world)
    my_groups = range(g_lo, g_hi)
    counts = [c * args.batch_size for c in dd.shard_counts(groups, world)]
    for img_idx, ref_img in enumerate(loader):
        logger.info(f"Inference for image {args.start_idx + img_idx}")
        fname = str(picks[img_idx]).zfill(5)
        ref_img = ref_img.to(device)
        os.makedirs(os.path.join(out_path, 'recon_paths', fname), exist_ok=True)
        os.makedirs(os.path.join(out_path, 'recon_paths_y', fname), exist_ok=True)

        if embedder is not None and hasattr(cond_method, 'guid_image_emb'):
            guid = ref_img
            if args.guid_image is not None:
                from PIL import Image
                guid = to_minus1_1(Image.open(args.guid_image).convert('RGB')).unsqueeze(0).to(device)
            with torch.no_grad():
                cond_method.guid_image_emb = embedder(guid).unsqueeze(0)        # [1, n_guid = 1, D]
        fkw = {}
        this_sample_fn = sample_fn
        if op_name == 'inpainting':
            mask = mask_gen(ref_img)[:, 0, :, :].unsqueeze(dim=0).contiguous()
            fkw = {'mask': mask}
            this_sample_fn = partial(sample_fn, measurement_cond_fn=partial(cond_method.conditioning, mask=mask, l1=args.l1),
                                     mask=mask)
        if world > 1:       # every rank must see the same measurement: rank 0 draws the noise
            gen_state = torch.random.get_rng_state()
        with torch.no_grad():
            y = operator.forward(ref_img, **fkw)
            y_n = noiser(y).contiguous()
        if world > 1:
            import torch.distributed as dist
            dist.broadcast(y_n, src=0)
            torch.random.set_rng_state(gen_state)
        C, H, W = ref_img.shape[1:]
        if rank == 0:
            imsave(os.path.join(out_path, 'input', fname + '.png'), clear_color(y_n))
            imsave(os.path.join(out_path, 'label', fname + '.png'), clear_color(ref_img))

        distances, finals = [], []
        for g in my_groups:
            x_start = torch.randn((args.batch_size, C, H, W), device=device).requires_grad_()
            sample = this_sample_fn(x_start=x_start, measurement=y_n, record=False, save_root=out_path)
            if isinstance(sample, tuple):       # ttc_ddim hands back (particles, distances) (reference :707)
                sample = sample[0]
            with torch.no_grad():
                y_space = operator.forward(sample, **fkw)                      # for the PNGs
                handle = operator.hip_handle_for(fkw['mask']) if op_name == 'inpainting' else operator.hip_handle(sample)
                dist_g = handle.score(sample, y_n)                             # ||y - A(x_p)||_2 per particle (HIP)
            distances.append(dist_g)
            finals.append(sample)
            for i in range(len(sample)):
                path_idx = args.path_start_idx + g * args.batch_size + i
                psnr = compute_psnr(ref_img, sample[i].unsqueeze(0))
                logger.info(f"Path#{path_idx + 1} | Method:{diffusion_config['sampler']} / PSNR: {float(psnr):.4f} / "
                            f"distance: {float(dist_g[i]):.4f}")
                imsave(os.path.join(out_path, 'recon_paths', fname, f'path#{path_idx + 1}.png'), clear_color(sample[i].unsqueeze(0)))
                imsave(os.path.join(out_path, 'recon_paths_y', fname, f'path#{path_idx + 1}_y_space.png'),
                       clear_color(y_space[i].unsqueeze(0)))
        # best-of-N over every particle of every rank: argmin of the final measurement distance.  Every rank enters
        # (a rank without groups contributes an empty shard and still receives the winner).
        scores_local = torch.cat(distances) if distances else torch.empty(0, device=device)
        particles_local = torch.cat(finals) if finals else torch.empty((0, C, H, W), device=device)
        winner, best, all_d = dd.global_best_of_n(scores_local, particles_local, counts)
        if rank == 0:
            logger.info(f"best-of-{all_d.numel()} = path#{args.path_start_idx + best + 1} | PSNR: "
                        f"{float(compute_psnr(ref_img, winner)):.4f} | distance: {float(all_d[best]):.4f}")
            imsave(os.path.join(out_path, 'best_of_n', fname + '.png'), clear_color(winner))
            np.save(os.path.join(out_path, f'{fname}_pathwise_distances.npy'), all_d.cpu().numpy())
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

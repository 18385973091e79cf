"""Denoiser: the ADM / guided-diffusion UNet the reference drives (guided_diffusion/unet.py:25-91, 467-734).

OUT OF SCOPE for hand-written kernels: the UNet runs through PyTorch-ROCm (MIOpen /
hipBLASLt) as a black box `model(x[N,3,H,W], t[1 or N]) -> [N, 6, H, W]`, and its VJP through torch.autograd.
This is a compact re-statement of the public architecture with the reference's `create_model(**model_yaml)`
signature and the same parameter names, so `models/ffhq_10m.pt` / `imagenet256.pt` load with
`load_state_dict` unchanged; a missing checkpoint falls back to random initialisation exactly as the reference
does (unet.py:87-90).  Attention uses torch's scaled_dot_product_attention (same maths as the legacy einsum path).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

NUM_CLASSES = 1000


class GroupNorm32(nn.GroupNorm):
    def forward(self, x):
        return super().forward(x.float()).type(x.dtype)


def zero_module(m):
    for p in m.parameters():
        p.detach().zero_()
    return m


def timestep_embedding(timesteps, dim, max_period=10000):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32, device=timesteps.device) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


class TimestepEmbedSequential(nn.Sequential):
    def forward(self, x, emb):
        for layer in self:
            x = layer(x, emb) if isinstance(layer, ResBlock) else layer(x)
        return x


class Upsample(nn.Module):
    def __init__(self, channels, use_conv, out_channels=None):
        super().__init__()
        self.use_conv = use_conv
        if use_conv:
            self.conv = nn.Conv2d(channels, out_channels or channels, 3, padding=1)

    def forward(self, x):
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        return self.conv(x) if self.use_conv else x


class Downsample(nn.Module):
    def __init__(self, channels, use_conv, out_channels=None):
        super().__init__()
        self.op = nn.Conv2d(channels, out_channels or channels, 3, stride=2, padding=1) if use_conv \
            else nn.AvgPool2d(kernel_size=2, stride=2)

    def forward(self, x):
        return self.op(x)


class ResBlock(nn.Module):
    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_scale_shift_norm=False,
                 up=False, down=False):
        super().__init__()
        out_channels = out_channels or channels
        self.use_scale_shift_norm = use_scale_shift_norm
        self.updown = up or down
        self.in_layers = nn.Sequential(GroupNorm32(32, channels), nn.SiLU(), nn.Conv2d(channels, out_channels, 3, padding=1))
        if up:
            self.h_upd, self.x_upd = Upsample(channels, False), Upsample(channels, False)
        elif down:
            self.h_upd, self.x_upd = Downsample(channels, False), Downsample(channels, False)
        else:
            self.h_upd = self.x_upd = nn.Identity()
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, 2 * out_channels if use_scale_shift_norm else out_channels))
        self.out_layers = nn.Sequential(GroupNorm32(32, out_channels), nn.SiLU(), nn.Dropout(p=dropout),
                                        zero_module(nn.Conv2d(out_channels, out_channels, 3, padding=1)))
        self.skip_connection = nn.Identity() if out_channels == channels else nn.Conv2d(channels, out_channels, 1)

    def forward(self, x, emb):
        if self.updown:
            h = self.in_layers[1](self.in_layers[0](x))
            h, x = self.h_upd(h), self.x_upd(x)
            h = self.in_layers[2](h)
        else:
            h = self.in_layers(x)
        e = self.emb_layers(emb).type(h.dtype)[..., None, None]
        if self.use_scale_shift_norm:
            scale, shift = torch.chunk(e, 2, dim=1)
            h = self.out_layers[0](h) * (1 + scale) + shift
            h = self.out_layers[3](self.out_layers[2](self.out_layers[1](h)))
        else:
            h = self.out_layers(h + e)
        return self.skip_connection(x) + h


class AttentionBlock(nn.Module):
    def __init__(self, channels, num_heads=1, num_head_channels=-1, use_new_attention_order=False):
        super().__init__()
        self.num_heads = num_heads if num_head_channels == -1 else channels // num_head_channels
        self.new_order = use_new_attention_order
        self.norm = GroupNorm32(32, channels)
        self.qkv = nn.Conv1d(channels, channels * 3, 1)
        self.proj_out = zero_module(nn.Conv1d(channels, channels, 1))

    def forward(self, x):
        b, c = x.shape[:2]
        flat = x.reshape(b, c, -1)
        qkv = self.qkv(self.norm(flat))
        t = qkv.shape[-1]
        hd = c // self.num_heads
        if self.new_order:          # QKVAttention: [q | k | v] blocks, heads inside
            q, k, v = (u.reshape(b, self.num_heads, hd, t) for u in qkv.chunk(3, dim=1))
        else:                       # QKVAttentionLegacy: per head [q, k, v]
            q, k, v = qkv.reshape(b, self.num_heads, 3 * hd, t).split(hd, dim=2)
        a = F.scaled_dot_product_attention(q.transpose(-1, -2), k.transpose(-1, -2), v.transpose(-1, -2))
        h = self.proj_out(a.transpose(-1, -2).reshape(b, c, t))
        return (flat + h).reshape(x.shape)


class UNetModel(nn.Module):
    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0, channel_mult=(1, 2, 4, 8), num_classes=None, use_checkpoint=False, use_fp16=False,
                 num_heads=1, num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=False,
                 resblock_updown=False, use_new_attention_order=False):
        super().__init__()
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads
        self.model_channels, self.num_classes = model_channels, num_classes
        self.dtype = torch.float16 if use_fp16 else torch.float32
        emb = model_channels * 4
        self.time_embed = nn.Sequential(nn.Linear(model_channels, emb), nn.SiLU(), nn.Linear(emb, emb))
        if num_classes is not None:
            self.label_emb = nn.Embedding(num_classes, emb)

        def res(cin, cout=None, **kw):
            return ResBlock(cin, emb, dropout, out_channels=cout, use_scale_shift_norm=use_scale_shift_norm, **kw)

        def attn(ch, heads):
            return AttentionBlock(ch, heads, num_head_channels, use_new_attention_order)

        ch = input_ch = int(channel_mult[0] * model_channels)
        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(nn.Conv2d(in_channels, ch, 3, padding=1))])
        chans, ds = [ch], 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [res(ch, int(mult * model_channels))]
                ch = int(mult * model_channels)
                if ds in attention_resolutions:
                    layers.append(attn(ch, num_heads))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(
                    res(ch, ch, down=True) if resblock_updown else Downsample(ch, True, ch)))
                chans.append(ch)
                ds *= 2
        self.middle_block = TimestepEmbedSequential(res(ch), attn(ch, num_heads), res(ch))
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                layers = [res(ch + chans.pop(), int(model_channels * mult))]
                ch = int(model_channels * mult)
                if ds in attention_resolutions:
                    layers.append(attn(ch, num_heads_upsample))
                if level and i == num_res_blocks:
                    layers.append(res(ch, ch, up=True) if resblock_updown else Upsample(ch, True, ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(GroupNorm32(32, ch), nn.SiLU(), zero_module(nn.Conv2d(input_ch, out_channels, 3, padding=1)))

    def forward(self, x, timesteps, y=None):
        emb = self.time_embed(timestep_embedding(timesteps, self.model_channels))
        if self.num_classes is not None:
            emb = emb + self.label_emb(y)
        hs, h = [], x.type(self.dtype)
        for module in self.input_blocks:
            h = module(h, emb)
            hs.append(h)
        h = self.middle_block(h, emb)
        for module in self.output_blocks:
            h = module(torch.cat([h, hs.pop()], dim=1), emb)
        return self.out(h.type(x.dtype))


def create_model(image_size, num_channels, num_res_blocks, channel_mult="", learn_sigma=False, class_cond=False,
                 use_checkpoint=False, attention_resolutions="16", num_heads=1, num_head_channels=-1,
                 num_heads_upsample=-1, use_scale_shift_norm=False, dropout=0, resblock_updown=False, use_fp16=False,
                 use_new_attention_order=False, model_path=''):
    """reference unet.py:25-91"""
    if channel_mult == "":
        table = {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4)}
        if image_size not in table:
            raise ValueError(f"unsupported image size: {image_size}")
        channel_mult = table[image_size]
    else:
        channel_mult = tuple(int(m) for m in channel_mult.split(","))
    if isinstance(attention_resolutions, int):
        attention_ds = [image_size // attention_resolutions]
    elif isinstance(attention_resolutions, str):
        attention_ds = [image_size // int(r) for r in attention_resolutions.split(",")]
    else:
        raise NotImplementedError
    model = UNetModel(image_size=image_size, in_channels=3, model_channels=num_channels,
                      out_channels=(6 if learn_sigma else 3), num_res_blocks=num_res_blocks,
                      attention_resolutions=tuple(attention_ds), dropout=dropout, channel_mult=channel_mult,
                      num_classes=(NUM_CLASSES if class_cond else None), use_checkpoint=use_checkpoint,
                      use_fp16=use_fp16, num_heads=num_heads, num_head_channels=num_head_channels,
                      num_heads_upsample=num_heads_upsample, use_scale_shift_norm=use_scale_shift_norm,
                      resblock_updown=resblock_updown, use_new_attention_order=use_new_attention_order)
    try:   # weights only: nothing from the file is executed
        model.load_state_dict(torch.load(model_path, map_location='cpu', weights_only=True))
    except Exception as e:   # same behaviour as the reference: a missing checkpoint means random weights
        print(f"Got exception: {e} / Randomly initialize")
    return model

"""Synthetic HBM-resident test frames for bench.py and tools/sweep.py (torch on the GPU).

Distributions (SURVEY.md 8(d) plus a natural-video stand-in), all BGRA8 with alpha 255:
  uniform       every B,G,R byte uniform 0..255
  concentrated  bytes = 120 + U{0..15}: one tight chroma cluster, heavy LDS contention
  flat          the reference UI test's colour (R,G,B) = (0xC6,0xCC,0xD7): one bin for everything
  gradient      smooth horizontal/vertical/diagonal ramps + 2 LSB of noise: neighbouring pixels share
                or nearly share their chroma, like real video
"""


def make_rgb_frames(torch, n, w, h, dist, seed):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    if dist == "uniform":
        rgb = torch.randint(0, 256, (n, h, w, 4), dtype=torch.uint8, device="cuda", generator=g)
    elif dist == "concentrated":
        rgb = torch.randint(120, 136, (n, h, w, 4), dtype=torch.uint8, device="cuda", generator=g)
    elif dist == "flat":
        rgb = torch.empty((n, h, w, 4), dtype=torch.uint8, device="cuda")
        rgb[..., 0], rgb[..., 1], rgb[..., 2] = 0xD7, 0xCC, 0xC6
    elif dist == "gradient":
        xs = torch.arange(w, device="cuda").view(1, 1, w)
        ys = torch.arange(h, device="cuda").view(1, h, 1)
        noise = torch.randint(0, 4, (n, h, w, 3), dtype=torch.int32, device="cuda", generator=g)
        rgb = torch.empty((n, h, w, 4), dtype=torch.uint8, device="cuda")
        rgb[..., 0] = ((xs * 255 // w) + noise[..., 0]).clamp(0, 255).to(torch.uint8)
        rgb[..., 1] = ((ys * 255 // h) + noise[..., 1]).clamp(0, 255).to(torch.uint8)
        rgb[..., 2] = (((xs + ys) * 255 // (w + h)) + noise[..., 2]).clamp(0, 255).to(torch.uint8)
        del noise
    else:
        raise ValueError(dist)
    rgb[..., 3] = 255
    return rgb.view(n, h, w * 4)


def make_frames(torch, ctx, n, w, h, dist, seed, colorspace=2):
    """(rgb, yuv) device tensors [n, h, 4*w]; yuv through the product's own K0 kernel."""
    rgb = make_rgb_frames(torch, n, w, h, dist, seed)
    yuv = torch.empty_like(rgb)
    for i in range(n):
        ctx.convert_rgb_yuv_dev(rgb[i], 4 * w, yuv[i], 4 * w, w, h, colorspace, sync=False)
    ctx.sync()
    return rgb, yuv

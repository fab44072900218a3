"""Synthetic HBM-resident test frames for bench.py and tools/sweep.py (torch on the GPU).

Distributions (SURVEY.md 8(d) plus a natural-video stand-in), all BGRA8 with alpha 255:
  uniform       every B,G,R byte uniform 0..255
  concentrated  bytes = 120 + U{0..15}: one tight chroma cluster, heavy LDS contention
  flat          the reference UI test's colour (R,G,B) = (0xC6,0xCC,0xD7): one bin for everything
  ramp          horizontal/vertical/diagonal ramps + 2 LSB of noise.  R is a blend of B and G here, so
                every pixel's (U,V) falls on ONE line of ~200 bins: a stress test (all bins saturate, every
                wave sees a handful of bins), not a picture
  screen        stand-in for screen capture / graphics: a flat background, 300 flat rectangles in random
                colours and a sprinkle of single-pixel detail (~3 %): most waves see one or two colours
  natural       stand-in for real video: a smooth random colour field (64x36 random RGB control points,
                bicubic up-sampling) + 2 LSB of sensor-like noise: neighbouring pixels share or nearly
                share their chroma, the frame as a whole covers a wide 2-D region of the CbCr plane
"""


def make_rgb_frames(torch, n, w, h, dist, seed, device="cuda"):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    if dist == "uniform":
        rgb = torch.randint(0, 256, (n, h, w, 4), dtype=torch.uint8, device=device, generator=g)
    elif dist == "concentrated":
        rgb = torch.randint(120, 136, (n, h, w, 4), dtype=torch.uint8, device=device, generator=g)
    elif dist == "flat":
        rgb = torch.empty((n, h, w, 4), dtype=torch.uint8, device=device)
        rgb[..., 0], rgb[..., 1], rgb[..., 2] = 0xD7, 0xCC, 0xC6
    elif dist == "screen":
        rgb = torch.empty((n, h, w, 4), dtype=torch.uint8, device=device)
        for i in range(n):
            img = torch.empty((h, w, 3), dtype=torch.uint8, device=device)
            img[...] = torch.randint(0, 256, (3,), dtype=torch.uint8, device=device, generator=g)
            cpu = torch.Generator()
            cpu.manual_seed(seed * 1000 + i)
            for _ in range(300):
                x0, y0 = int(torch.randint(0, w - 8, (1,), generator=cpu)), int(torch.randint(0, h - 8, (1,), generator=cpu))
                ww, hh = int(torch.randint(8, 600, (1,), generator=cpu)), int(torch.randint(8, 300, (1,), generator=cpu))
                img[y0:y0 + hh, x0:x0 + ww] = torch.randint(0, 256, (3,), dtype=torch.uint8, device=device, generator=g)
            detail = torch.rand((h, w), device=device, generator=g) < 0.03
            img[detail] = torch.randint(0, 256, (int(detail.sum()), 3), dtype=torch.uint8, device=device, generator=g)
            rgb[i, :, :, :3] = img
    elif dist == "natural":
        import torch.nn.functional as F

        rgb = torch.empty((n, h, w, 4), dtype=torch.uint8, device=device)
        for i in range(n):
            ctrl = torch.rand((1, 3, 36, 64), device=device, generator=g) * 255.0
            img = F.interpolate(ctrl, size=(h, w), mode="bicubic", align_corners=False)
            img = img + torch.randint(0, 4, (1, 3, h, w), device=device, generator=g).float() - 1.5
            rgb[i, :, :, :3] = img.clamp(0, 255).round().to(torch.uint8)[0].permute(1, 2, 0)
    elif dist == "ramp":
        xs = torch.arange(w, device=device).view(1, 1, w)
        ys = torch.arange(h, device=device).view(1, h, 1)
        noise = torch.randint(0, 4, (n, h, w, 3), dtype=torch.int32, device=device, generator=g)
        rgb = torch.empty((n, h, w, 4), dtype=torch.uint8, device=device)
        rgb[..., 0] = ((xs * 255 // w) + noise[..., 0]).clamp(0, 255).to(torch.uint8)
        rgb[..., 1] = ((ys * 255 // h) + noise[..., 1]).clamp(0, 255).to(torch.uint8)
        rgb[..., 2] = (((xs + ys) * 255 // (w + h)) + noise[..., 2]).clamp(0, 255).to(torch.uint8)
        del noise
    else:
        raise ValueError(dist)
    rgb[..., 3] = 255
    return rgb.view(n, h, w * 4)


def make_frames(torch, ctx, n, w, h, dist, seed, colorspace=2, device="cuda"):
    """(rgb, yuv) device tensors [n, h, 4*w]; yuv through the product's own K0 kernel."""
    rgb = make_rgb_frames(torch, n, w, h, dist, seed, device)
    yuv = torch.empty_like(rgb)
    for i in range(n):
        ctx.convert_rgb_yuv_dev(rgb[i], 4 * w, yuv[i], 4 * w, w, h, colorspace, sync=False)
    ctx.sync()
    return rgb, yuv


def make_frame_tiles(torch, ctx, n, w, h, r0, r1, seed, colorspace=2, device="cuda"):
    """Rows [r0, r1) of `n` uniform w x h frames that every rank generates from the SAME seed: frame f is drawn whole
    from seed + f (one frame at a time, 133 MB at 7680x4320) and only the rank's rows are kept, so N ranks hold N
    tiles of ONE frame set whatever N is (BASELINE config 4: one 7680x4320 batch in row tiles)."""
    g = torch.Generator(device=device)
    th = r1 - r0
    rgb = torch.empty((n, th, w * 4), dtype=torch.uint8, device=device)
    for f in range(n):
        g.manual_seed(seed + f)
        full = torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device=device, generator=g)
        full[..., 3] = 255
        rgb[f] = full[r0:r1].reshape(th, w * 4)
        del full
    yuv = torch.empty_like(rgb)
    for f in range(n):
        ctx.convert_rgb_yuv_dev(rgb[f], 4 * w, yuv[f], 4 * w, w, th, colorspace, sync=False)
    ctx.sync()
    return rgb, yuv


def punch_alpha_holes(torch, rgb, w):
    """alpha = 0 on a checker of 16x16 blocks (the waveform's and histogram's `if (!a) continue`)."""
    n, h = rgb.shape[0], rgb.shape[1]
    v = rgb.view(n, h, -1, 4)[:, :, :w]
    ys = (torch.arange(h, device=rgb.device) // 16).view(1, h, 1)
    xs = (torch.arange(w, device=rgb.device) // 16).view(1, 1, w)
    v[..., 3] = torch.where(((xs + ys) & 1).bool(), 0, 255).to(torch.uint8).expand(n, h, w)
    return rgb


def pad_pitch(torch, planes, w, extra=256):
    """the same planes with linesize = 4 W + `extra` (what a padded D3D11 staging surface looks like)."""
    n, h = planes.shape[0], planes.shape[1]
    out = torch.zeros((n, h, 4 * w + extra), dtype=torch.uint8, device=planes.device)
    out[:, :, : 4 * w] = planes[:, :, : 4 * w]
    return out

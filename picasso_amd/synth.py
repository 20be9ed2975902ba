"""Seeded synthetic DNA-PAINT movies.

The reference's simulator (picasso/simulate.py:357-490) draws every photon with
``multivariate_normal`` + ``histogram2d`` and cannot produce 10k x 512 x 512
frames.  This generator follows its *model* — pixel-integrated Gaussian PSF,
Poisson shot noise, baseline offset, uint16 clip (picasso/simulate.py:138-154) —
with torch ops so that the movie can be produced directly in HBM.  It is a
test/bench data source, not part of the hot path.

Emitters sit on a jittered 16-px grid (centres >= 8 px apart) so their boxes
never overlap; photons ~ U(photons), sigma_x = sigma_y ~ U(sigma), background
~ U(bg) photons/px per frame, camera = {Baseline, Sensitivity 1, Gain 1}.
"""
from __future__ import annotations

import math

import numpy as np
import torch

DEFAULT_SEED = 20261001


def _to_uint16(t: torch.Tensor) -> torch.Tensor:
    t = t.clamp_(0, 65535).to(torch.int32)
    t = torch.where(t > 32767, t - 65536, t).to(torch.int16)
    return t.view(torch.uint16)


@torch.no_grad()
def simulate_movie(n_frames: int, height: int = 512, width: int = 512, emitters_per_frame: int = 100,
                   seed: int = DEFAULT_SEED, device="cuda", photons=(2000.0, 8000.0), sigma=(0.9, 1.4),
                   bg=(10.0, 30.0), baseline: float = 100.0, astigmatic: bool = False,
                   chunk_frames: int = 128, return_truth: bool = False):
    """-> uint16 tensor (n_frames, height, width) on `device` [, truth dict]."""
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed))
    cell = 16
    ncy, ncx = (height - 16) // cell, (width - 16) // cell
    ncells = ncy * ncx
    k = min(int(emitters_per_frame), ncells)
    out = torch.empty((n_frames, height, width), dtype=torch.uint16, device=dev)
    P = 15                                    # patch side: +-7 px covers > 5 sigma at sigma 1.4
    half = P // 2
    off = torch.arange(P, device=dev, dtype=torch.float32) - half
    truth = {"frame": [], "x": [], "y": [], "photons": [], "sx": [], "sy": [], "bg": []}
    inv_sqrt2 = 1.0 / math.sqrt(2.0)
    for f0 in range(0, n_frames, chunk_frames):
        nf = min(chunk_frames, n_frames - f0)
        bgv = torch.empty(nf, device=dev).uniform_(bg[0], bg[1], generator=gen)
        rate = bgv.view(nf, 1, 1).expand(nf, height, width).contiguous()
        if k > 0:
            cells = torch.rand((nf, ncells), device=dev, generator=gen).argsort(dim=1)[:, :k]
            cy = torch.div(cells, ncx, rounding_mode="floor")
            cx = cells - cy * ncx
            x0 = 8 + cx * cell + 4 + torch.empty((nf, k), device=dev).uniform_(0, 8, generator=gen)
            y0 = 8 + cy * cell + 4 + torch.empty((nf, k), device=dev).uniform_(0, 8, generator=gen)
            ph = torch.empty((nf, k), device=dev).uniform_(photons[0], photons[1], generator=gen)
            sx = torch.empty((nf, k), device=dev).uniform_(sigma[0], sigma[1], generator=gen)
            sy = torch.empty((nf, k), device=dev).uniform_(sigma[0], sigma[1], generator=gen) if astigmatic else sx
            ix = torch.floor(x0).to(torch.int64)     # pixel holding the centre (pixel i spans [i-0.5, i+0.5))
            iy = torch.floor(y0).to(torch.int64)
            # pixel centres are integers: pixel i integrates [i-0.5, i+0.5]
            px = ix.unsqueeze(-1) + off.to(torch.int64)            # (nf, k, P)
            py = iy.unsqueeze(-1) + off.to(torch.int64)
            dx = px.to(torch.float32) - x0.unsqueeze(-1)
            dy = py.to(torch.float32) - y0.unsqueeze(-1)
            ex = 0.5 * (torch.erf((dx + 0.5) * inv_sqrt2 / sx.unsqueeze(-1)) - torch.erf((dx - 0.5) * inv_sqrt2 / sx.unsqueeze(-1)))
            ey = 0.5 * (torch.erf((dy + 0.5) * inv_sqrt2 / sy.unsqueeze(-1)) - torch.erf((dy - 0.5) * inv_sqrt2 / sy.unsqueeze(-1)))
            patch = ph.view(nf, k, 1, 1) * ey.unsqueeze(-1) * ex.unsqueeze(-2)        # (nf, k, P, P)
            fidx = torch.arange(nf, device=dev).view(nf, 1, 1, 1).expand(nf, k, P, P)
            yidx = py.clamp(0, height - 1).unsqueeze(-1).expand(nf, k, P, P)
            xidx = px.clamp(0, width - 1).unsqueeze(-2).expand(nf, k, P, P)
            rate.index_put_((fidx, yidx, xidx), patch, accumulate=True)
            if return_truth:
                truth["frame"].append((torch.arange(nf, device=dev).view(nf, 1).expand(nf, k) + f0).reshape(-1).cpu())
                for name, val in (("x", x0), ("y", y0), ("photons", ph), ("sx", sx), ("sy", sy)):
                    truth[name].append(val.reshape(-1).cpu())
                truth["bg"].append(bgv.view(nf, 1).expand(nf, k).reshape(-1).cpu())
        counts = torch.poisson(rate, generator=gen) + baseline
        out[f0:f0 + nf] = _to_uint16(counts)
        del rate, counts
    if return_truth:
        return out, {kk: torch.cat(v).numpy() if v else np.zeros(0) for kk, v in truth.items()}
    return out

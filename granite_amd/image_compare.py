"""PSNR gate over rendered frames: same arguments, arithmetic and exit codes as Granite's tools/image_compare.cpp.

    python -m granite_amd.image_compare A B [--threshold dB] [--diff out.png]

A and B are two images (.gtx or .png, RGBA8 UNORM / SRGB) or two directories compared entry by entry in sorted order
(image_compare.cpp:149-199).  PSNR = 10 log10(255^2 * 3 W H / sum of squared RGB byte differences), alpha ignored
(:87-116); identical images give +inf, like the reference's division by a zero error energy.  Exit code 1 when any PSNR
is below --threshold, an input cannot be loaded, or the two folders differ in size."""
from __future__ import annotations

import argparse
import math
import os
import sys

import numpy as np

from . import gtx, png

FORMAT_R8G8B8A8_UNORM, FORMAT_R8G8B8A8_SRGB = 37, 43


def load_image(path: str):
    """(format, (H, W, 4) uint8) or None when the file cannot be loaded (load_texture_from_file's empty texture)."""
    try:
        if path.lower().endswith(".png"):
            return FORMAT_R8G8B8A8_SRGB, png.read_png(path)   # stb loads PNG as sRGB (texture_files.cpp)
        img = gtx.read(path)
        if img.info.format not in (FORMAT_R8G8B8A8_UNORM, FORMAT_R8G8B8A8_SRGB):
            return img.info.format, None
        return img.info.format, img.level(0)[0]
    except (OSError, ValueError, gtx.GtxError) as e:
        print(f"[ERROR]: {e}", file=sys.stderr)
        return None


def compare_images(a, b) -> float:
    (fa, ia), (fb, ib) = a, b
    if fa != fb:
        print("[ERROR]: Format mismatch.", file=sys.stderr)
        return 0.0
    if ia is None or ib is None:
        print("[ERROR]: Unsupported format.", file=sys.stderr)
        return 0.0
    if ia.shape != ib.shape:
        print("[ERROR]: Dimension mismatch.", file=sys.stderr)
        return 0.0
    h, w = ia.shape[:2]
    diff = ia[..., :3].astype(np.int64) - ib[..., :3].astype(np.int64)
    error_energy = float((diff * diff).sum())
    peak_energy = 255.0 * 255.0 * w * h * 3.0
    return math.inf if error_energy == 0.0 else 10.0 * math.log10(peak_energy / error_energy)


def diff_image(a, b) -> np.ndarray:
    """save_diff_image (:40-85): min(16 * (a - b), 255) per colour channel stored as a byte, alpha 255.  The reference
    narrows the possibly negative int to uint8_t, i.e. modulo 256; reproduced."""
    d = (a[1][..., :3].astype(np.int32) - b[1][..., :3].astype(np.int32)) * 16
    out = np.full(a[1].shape, 255, np.uint8)
    out[..., :3] = (np.minimum(d, 255) & 255).astype(np.uint8)
    return out


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("inputs", nargs="*")
    ap.add_argument("--threshold", type=float, default=-1.0)
    ap.add_argument("--diff", default="")
    args = ap.parse_args(argv)
    if len(args.inputs) != 2:
        print("[ERROR]: Need two inputs.", file=sys.stderr)
        return 1
    pa, pb = args.inputs
    if os.path.isdir(pa) and os.path.isdir(pb):
        la = sorted(os.path.join(pa, n) for n in os.listdir(pa))
        lb = sorted(os.path.join(pb, n) for n in os.listdir(pb))
        if len(la) != len(lb):
            print("[ERROR]: Folder size is not identical.", file=sys.stderr)
            return 1
        for fa, fb in zip(la, lb):
            a, b = load_image(fa), load_image(fb)
            if a is None or b is None:
                continue
            psnr = compare_images(a, b)
            print(f"{fa} | {fb} | PSNR: {psnr:.0f} dB")
            if 0.0 <= args.threshold and psnr < args.threshold:
                print("[ERROR]: PSNR is too low, failure!", file=sys.stderr)
                return 1
        return 0
    a = load_image(pa)
    if a is None:
        print(f"[ERROR]: Failed to load texture: {pa}", file=sys.stderr)
        return 1
    b = load_image(pb)
    if b is None:
        print(f"[ERROR]: Failed to load texture: {pb}", file=sys.stderr)
        return 1
    if args.diff:
        if a[0] != b[0]:
            print("[ERROR]: Format mismatch.", file=sys.stderr)
        elif a[1] is None or b[1] is None or a[1].shape != b[1].shape:
            print("[ERROR]: Unsupported format.", file=sys.stderr)
        else:
            png.write_png(args.diff, diff_image(a, b))
    psnr = compare_images(a, b)
    print(f"PSNR: {psnr:.0f} dB")
    if 0.0 <= args.threshold and psnr < args.threshold:
        print("[ERROR]: PSNR is too low, failure!", file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())

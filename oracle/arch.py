"""EfficientNet architecture tables (oracle side).

Restates model/modules/efficient_net_custom_utils.py:83-126 (round_filters / round_repeats),
:457-479 (compound coefficients), :482-528 (the 7-stage base block table), and the *static* SAME
padding of Conv2dStaticSamePadding (:248-276), which is frozen at construction time for the
network's NOMINAL resolution (260 for B2, 456 for B5) and therefore is NOT TF-SAME at the real
input size.
"""
import math
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

# (width, depth, nominal resolution, dropout) -- efficient_net_custom_utils.py:457-479
COEFFS = {
    "efficientnet-b0": (1.0, 1.0, 224, 0.2),
    "efficientnet-b1": (1.0, 1.1, 240, 0.2),
    "efficientnet-b2": (1.1, 1.2, 260, 0.3),
    "efficientnet-b3": (1.2, 1.4, 300, 0.3),
    "efficientnet-b4": (1.4, 1.8, 380, 0.4),
    "efficientnet-b5": (1.6, 2.2, 456, 0.4),
}

# (repeats, kernel, stride, expand, in, out, se_ratio) -- efficient_net_custom_utils.py:502-510
BASE_STAGES = [
    (1, 3, 1, 1, 32, 16, 0.25),
    (2, 3, 2, 6, 16, 24, 0.25),
    (2, 5, 2, 6, 24, 40, 0.25),
    (3, 3, 2, 6, 40, 80, 0.25),
    (3, 5, 1, 6, 80, 112, 0.25),
    (4, 5, 2, 6, 112, 192, 0.25),
    (1, 3, 1, 6, 192, 320, 0.25),
]

BN_EPS = 1e-3          # efficient_net_custom_utils.py:521
BN_MOMENTUM = 0.01     # 1 - 0.99, efficientnet_custom.py:53
DROP_CONNECT = 0.2     # efficient_net_custom_utils.py:483


def round_filters(filters: int, width: float, divisor: int = 8) -> int:
    """efficient_net_custom_utils.py:83-108."""
    if not width:
        return filters
    f = filters * width
    new_f = max(divisor, int(f + divisor / 2) // divisor * divisor)
    if new_f < 0.9 * f:
        new_f += divisor
    return int(new_f)


def round_repeats(repeats: int, depth: float) -> int:
    """efficient_net_custom_utils.py:111-126."""
    if not depth:
        return repeats
    return int(math.ceil(depth * repeats))


def same_pad(size_hw: Tuple[int, int], k: int, s: int) -> Tuple[int, int, int, int]:
    """Static SAME padding (left, right, top, bottom) for a conv of kernel k, stride s whose
    *nominal* input is size_hw -- efficient_net_custom_utils.py:262-272."""
    ih, iw = size_hw
    oh, ow = math.ceil(ih / s), math.ceil(iw / s)
    pad_h = max((oh - 1) * s + (k - 1) + 1 - ih, 0)
    pad_w = max((ow - 1) * s + (k - 1) + 1 - iw, 0)
    return (pad_w // 2, pad_w - pad_w // 2, pad_h // 2, pad_h - pad_h // 2)


def out_size(size_hw, s):
    """efficient_net_custom_utils.py:176-194 (calculate_output_image_size)."""
    return (int(math.ceil(size_hw[0] / s)), int(math.ceil(size_hw[1] / s)))


def conv_out(i: int, pad_lo: int, pad_hi: int, k: int, s: int) -> int:
    """Real output extent of F.conv2d(padding=0) after the static ZeroPad2d."""
    return (i + pad_lo + pad_hi - k) // s + 1


@dataclass
class Block:
    idx: int
    expand: int
    k: int
    s: int
    cin: int
    cexp: int
    cout: int
    cse: int
    pad: Tuple[int, int, int, int]   # (left, right, top, bottom) of the depthwise conv
    skip: bool
    drop_rate: float = 0.0           # DROP_CONNECT * idx / n_blocks (efficientnet_custom.py:277-279)


@dataclass
class Arch:
    name: str
    stem_out: int
    stem_pad: Tuple[int, int, int, int]
    head_in: int
    head_out: int
    dropout: float
    blocks: List[Block] = field(default_factory=list)


def build_arch(name: str, width: Optional[float] = None, depth: Optional[float] = None,
               image_size: Optional[int] = None) -> Arch:
    """Block list exactly as EfficientNet.__init__ builds it (efficientnet_custom.py:158-214).
    width/depth/image_size override the compound coefficients like ``from_name(**override)``."""
    w, d, res, p = COEFFS[name]
    if width is not None:
        w = width
    if depth is not None:
        d = depth
    if image_size is not None:
        res = image_size
    size = (res, res)
    stem_out = round_filters(32, w)
    stem_pad = same_pad(size, 3, 2)
    size = out_size(size, 2)
    blocks: List[Block] = []
    for (r, k, s, e, i, o, se) in BASE_STAGES:
        cin, cout, reps = round_filters(i, w), round_filters(o, w), round_repeats(r, d)
        for rep in range(reps):
            b_in = cin if rep == 0 else cout
            b_s = s if rep == 0 else 1
            blocks.append(Block(
                idx=len(blocks), expand=e, k=k, s=b_s, cin=b_in, cexp=b_in * e, cout=cout,
                cse=max(1, int(b_in * se)), pad=same_pad(size, k, b_s),
                skip=(b_s == 1 and b_in == cout)))
            if rep == 0:
                size = out_size(size, s)
    for b in blocks:
        b.drop_rate = DROP_CONNECT * float(b.idx) / len(blocks)
    return Arch(name=name, stem_out=stem_out, stem_pad=stem_pad, head_in=blocks[-1].cout,
                head_out=round_filters(1280, w), dropout=p, blocks=blocks)


def spatial_chain(arch: Arch, h: int, w: int):
    """Real (H, W) after the stem and after every block for a real input h x w."""
    l, r, t, b = arch.stem_pad
    h, w = conv_out(h, t, b, 3, 2), conv_out(w, l, r, 3, 2)
    out = [(h, w)]
    for blk in arch.blocks:
        l, r, t, b = blk.pad
        h, w = conv_out(h, t, b, blk.k, blk.s), conv_out(w, l, r, blk.k, blk.s)
        out.append((h, w))
    return out

"""On-wire container for one image coded at several mask levels.

The reference has no bitstream format: compress() returns Python lists of byte strings and the harness only sums their
lengths (training/step.py:357-365).  This module defines one, built around what is shared between levels: the hyper-latent
string and the ten base-slice strings are stored once, each level with quality > 0 adds its ten enhancement strings.

Layout (little endian):

    magic  "PCB1"                      4 B
    version                            u8   (= 2)
    numeric contract id                u32  (pc_contract_id() of the encoder: include/pc_math.h)
    mask policy                        u8   (1 point-based-std, 2 two-levels, 3 three-levels-std)
    n_levels                           u16
    H, W  (original, un-padded size)   u32 u32
    zh, zw ("shape" of compress())     u16 u16
    quality[n_levels]                  f64 each
    len(z), len(base slice 0..9)       u32 x 11
    for every level with quality > 0:  len(enhancement slice 0..9)   u32 x 10
    payload: z, base 0..9, then the enhancement strings level by level, in the order of the length table

Progressive property: the prefix ending with base slice 9 decodes level quality 0; each level's segment is independent of the
other levels', so a reader wanting level l needs the header, the base segment and that level's segment only
(`unpack(..., levels=[l])`).  The header costs 24 + 8*n_levels + 44 + 40*n_coded_levels bytes per image, which the reference's bpp
(sum of string lengths) does not count; `payload_bytes()` returns the reference's figure.

Interchange: the strings are decodable only by a decoder built to the SAME numeric contract -- the decoder
re-derives mu / scale / mask from decoded data, so a build (or the reference's PyTorch path) that rounds one float differently
desynchronises rANS and decodes garbage without any error.  The header therefore carries the encoder's contract id and `unpack`
refuses a container of another contract (version-1 containers, which carry none, are refused as well).
"""
import struct

MAGIC = b"PCB1"
VERSION = 2
_HEAD = "<BIBHIIHH"      # version, contract id, mask policy, n_levels, H, W, zh, zw


def build_contract_id():
    """pc_contract_id() of the loaded libpcodec.so"""
    from ._lib import lib
    return int(lib().pc_contract_id())
MASK_POL = {"point-based-std": 1, "two-levels": 2, "three-levels-std": 3}
_MASK_POL_INV = {v: k for k, v in MASK_POL.items()}


class ContainerError(ValueError):
    pass


def pack(strings_per_level, shape, qualities, image_size, mask_pol="point-based-std", image_index=0, contract=None):
    """strings_per_level[l] = [y_strings, z_strings] as returned by compress()/compress_levels() for level l
    (y_strings[slice][image]); one container holds ONE image (image_index of the batch)."""
    qualities = [float(q) for q in qualities]
    if len(strings_per_level) != len(qualities) or not qualities:
        raise ContainerError("one [y_strings, z_strings] entry per level")
    if mask_pol not in MASK_POL:
        raise ContainerError(f"mask policy {mask_pol!r}")
    b = image_index
    y0, z0 = strings_per_level[0]
    if len(y0) < 10:
        raise ContainerError("ten base slices expected")
    z = bytes(z0[b])
    base = [bytes(y0[s][b]) for s in range(10)]
    enh = []
    for q, (ys, zs) in zip(qualities, strings_per_level):
        if bytes(zs[b]) != z or [bytes(ys[s][b]) for s in range(10)] != base:
            raise ContainerError("levels of one container must share the hyper-latent and base strings")
        if q > 0:
            if len(ys) < 20:
                raise ContainerError(f"level {q}: ten enhancement slices expected")
            enh.append([bytes(ys[s][b]) for s in range(10, 20)])
    H, W = int(image_size[0]), int(image_size[1])
    contract = build_contract_id() if contract is None else int(contract)
    head = MAGIC + struct.pack(_HEAD, VERSION, contract, MASK_POL[mask_pol], len(qualities), H, W, int(shape[0]), int(shape[1]))
    head += struct.pack(f"<{len(qualities)}d", *qualities)
    parts = [z] + base + [s for lv in enh for s in lv]
    if any(len(p) >= 1 << 32 for p in parts):
        raise ContainerError("string too long")
    head += struct.pack(f"<{len(parts)}I", *[len(p) for p in parts])
    return head + b"".join(parts)


def parse_header(buf):
    """-> dict(mask_pol, qualities, image_size, shape, lengths (z, base[10], enh[level][10]), header_bytes)."""
    if len(buf) < 20 or buf[:4] != MAGIC:
        raise ContainerError("not a PCB1 container")
    ver = buf[4]
    if ver != VERSION:
        raise ContainerError(f"unsupported version {ver}" + (" (version 1 carries no numeric-contract id: not decodable safely)" if ver == 1 else ""))
    if len(buf) < 4 + struct.calcsize(_HEAD):
        raise ContainerError("truncated header")
    ver, contract, mp, n_levels, H, W, zh, zw = struct.unpack_from(_HEAD, buf, 4)
    if mp not in _MASK_POL_INV or n_levels == 0:
        raise ContainerError("corrupt header")
    off = 4 + struct.calcsize(_HEAD)
    if len(buf) < off + 8 * n_levels:
        raise ContainerError("truncated header")
    qualities = list(struct.unpack_from(f"<{n_levels}d", buf, off))
    off += 8 * n_levels
    n_coded = sum(1 for q in qualities if q > 0)
    n_parts = 11 + 10 * n_coded
    if len(buf) < off + 4 * n_parts:
        raise ContainerError("truncated header")
    lens = list(struct.unpack_from(f"<{n_parts}I", buf, off))
    off += 4 * n_parts
    return {"contract": contract, "mask_pol": _MASK_POL_INV[mp], "qualities": qualities, "image_size": (H, W), "shape": (zh, zw),
            "z_len": lens[0], "base_lens": lens[1:11], "enh_lens": [lens[11 + 10 * i:21 + 10 * i] for i in range(n_coded)],
            "header_bytes": off}


def unpack(buf, levels=None, expect_contract=None):
    """-> (strings_per_level, shape, qualities, image_size, mask_pol) for the requested level indices (default: all), in the
    nesting compress() uses (y_strings[slice][image] with one image).  Only the bytes of the header, the base segment and the
    requested levels' segments are touched, so a truncated file still yields the levels it holds completely.  expect_contract: the
    numeric-contract id the decoder implements (default: the loaded library's); a container of another contract is refused
    (False: skip the check, e.g. to inspect lengths only)."""
    hd = parse_header(buf)
    expect = build_contract_id() if expect_contract is None else expect_contract
    if expect is not False and hd["contract"] != int(expect):
        raise ContainerError(f"container was coded under numeric contract 0x{hd['contract']:08x}, this decoder implements 0x{int(expect):08x}: "
                             "the streams are not interchangeable (DESIGN.md section 2)")
    qualities = hd["qualities"]
    want = list(range(len(qualities))) if levels is None else list(levels)
    off = hd["header_bytes"]

    def take(n):
        nonlocal off
        if off + n > len(buf):
            raise ContainerError("truncated payload")
        s = bytes(buf[off:off + n])
        off += n
        return s

    z = take(hd["z_len"])
    base = [take(n) for n in hd["base_lens"]]
    seg_start, pos, ci = {}, off, 0
    for lv, q in enumerate(qualities):
        if q > 0:
            seg_start[lv] = (pos, hd["enh_lens"][ci])
            pos += sum(hd["enh_lens"][ci])
            ci += 1
    out = []
    for lv in want:
        if lv < 0 or lv >= len(qualities):
            raise ContainerError(f"no level {lv}")
        ys = [[s] for s in base]
        if qualities[lv] > 0:
            off, lens = seg_start[lv]
            ys += [[take(n)] for n in lens]
        out.append([ys, [z]])
    return out, hd["shape"], [qualities[lv] for lv in want], hd["image_size"], hd["mask_pol"]


def payload_bytes(buf, level):
    """bytes the reference's bpp counts for one level (step.py:357-365): z + base + that level's enhancement strings."""
    hd = parse_header(buf)
    n = hd["z_len"] + sum(hd["base_lens"])
    q = hd["qualities"][level]
    if q > 0:
        ci = sum(1 for x in hd["qualities"][:level] if x > 0)
        n += sum(hd["enh_lens"][ci])
    return n

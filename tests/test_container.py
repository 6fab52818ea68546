"""The multi-level container (progressivecodec_amd/container.py): format round trips and error behaviour.  Host logic only."""
import os
import struct

import pytest

from progressivecodec_amd import container as ct


def _levels(seed=1):
    import random
    r = random.Random(seed)
    rb = lambda n: bytes(r.getrandbits(8) for _ in range(n))
    z = [rb(11)]
    base = [[rb(r.randrange(0, 40))] for _ in range(10)]
    enh = lambda: [[rb(r.randrange(0, 60))] for _ in range(10)]
    return [[base, z], [base + enh(), z], [base + enh(), z], [base, z]], [0, 0.5, 2, 0]


def test_round_trip_all_and_selected_levels():
    lv, q = _levels()
    buf = ct.pack(lv, (4, 6), q, (250, 380))
    out, shape, qq, size, mp = ct.unpack(buf)
    assert out == lv and tuple(shape) == (4, 6) and qq == q and size == (250, 380) and mp == "point-based-std"
    out2, _, q2, _, _ = ct.unpack(buf, [2, 0])
    assert out2 == [lv[2], lv[0]] and q2 == [2, 0]
    hd = ct.parse_header(buf)
    assert hd["header_bytes"] == 4 + 20 + 8 * 4 + 4 * (11 + 20) and hd["contract"] == ct.build_contract_id()
    assert len(buf) == hd["header_bytes"] + hd["z_len"] + sum(hd["base_lens"]) + sum(map(sum, hd["enh_lens"]))
    for l in range(4):
        ys, zs = lv[l]
        assert ct.payload_bytes(buf, l) == sum(len(s[0]) for s in ys) + len(zs[0])      # the reference's byte count, step.py:357-365


def test_prefix_decodes_the_levels_it_holds():
    lv, q = _levels(2)
    buf = ct.pack(lv, (1, 1), q, (64, 64))
    hd = ct.parse_header(buf)
    base_end = hd["header_bytes"] + hd["z_len"] + sum(hd["base_lens"])
    out, *_ = ct.unpack(buf[:base_end], [0, 3])
    assert out == [lv[0], lv[3]]
    with pytest.raises(ct.ContainerError):
        ct.unpack(buf[:base_end], [1])
    lvl1_end = base_end + sum(hd["enh_lens"][0])
    out, *_ = ct.unpack(buf[:lvl1_end], [1])
    assert out == [lv[1]]
    with pytest.raises(ct.ContainerError):
        ct.unpack(buf[:lvl1_end], [2])


def test_second_image_of_a_batch_and_two_levels_policy():
    z = [b"a", b"bb"]
    base = [[bytes([i]), bytes([i, i])] for i in range(10)]
    enh = [[bytes([9 - i]) * 3, bytes([9 - i]) * 4] for i in range(10)]
    buf = ct.pack([[base + enh, z]], (2, 3), [10], (128, 192), mask_pol="two-levels", image_index=1)
    out, shape, q, size, mp = ct.unpack(buf)
    assert out == [[[[s[1]] for s in base + enh], [z[1]]]] and mp == "two-levels" and q == [10.0]


def test_errors():
    lv, q = _levels(3)
    buf = ct.pack(lv, (4, 6), q, (250, 380))
    with pytest.raises(ct.ContainerError):
        ct.parse_header(b"XXXX" + buf[4:])
    with pytest.raises(ct.ContainerError):
        ct.parse_header(buf[:4] + bytes([9]) + buf[5:])            # unknown version
    with pytest.raises(ct.ContainerError):
        ct.parse_header(buf[:30])                                   # truncated header
    with pytest.raises(ct.ContainerError) as ei:
        ct.parse_header(buf[:4] + bytes([1]) + buf[5:])            # version 1: no contract id
    assert "contract" in str(ei.value)
    foreign = ct.pack(lv, (4, 6), q, (250, 380), contract=0x00010001)   # a stream coded under another contract revision
    with pytest.raises(ct.ContainerError) as ei:
        ct.unpack(foreign)
    assert "not interchangeable" in str(ei.value)
    assert ct.unpack(foreign, expect_contract=False)[0] == ct.unpack(buf)[0]          # inspection without the check
    assert ct.unpack(foreign, expect_contract=0x00010001)[0] == ct.unpack(buf)[0]
    with pytest.raises(ct.ContainerError):
        ct.unpack(buf, [7])
    with pytest.raises(ct.ContainerError):
        ct.pack(lv, (4, 6), q[:-1], (250, 380))
    other = [[[b"x"]] * 10, lv[0][1]]
    with pytest.raises(ct.ContainerError):
        ct.pack([lv[0], other], (4, 6), [0, 0], (250, 380))         # levels must share the base strings
    with pytest.raises(ct.ContainerError):
        ct.pack([lv[0]], (4, 6), [0.5], (250, 380))                 # quality > 0 needs enhancement strings

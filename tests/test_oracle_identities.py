"""CPU: self-consistency identities that pin the oracle where the reference has no golden vectors
 -- transforms, tables, motion compensation."""
import ctypes as C

import numpy as np
import pytest

from tests import tables
from tests.oracle_binding import lib as oracle_lib

T = tables.load()


def _idct(n, coef, variant, pred):
    L = oracle_lib()
    stride = 32
    dst = np.ascontiguousarray(np.full((n + 2) * stride, 0, np.uint8))
    dst.reshape(n + 2, stride)[1:n + 1, 8:8 + n] = pred
    c = np.ascontiguousarray(coef, dtype=np.int32)
    f = L.mobi_oracle_idct8 if n == 8 else L.mobi_oracle_idct4
    rc = f(c.ctypes.data, variant, dst.ctypes.data, dst.size, stride + 8, stride)
    return rc, dst.reshape(n + 2, stride)[1:n + 1, 8:8 + n].copy()


def test_reduced_idct_equals_full_on_zero_extended_input():
    """MD.cs picks IDCT{1,3,16}Px8 / IDCT1Px4 by the last scan index (:2939-2942, :2954-2955); they must be
    exact specialisations of the full transforms, which is what lets the GPU always run the full one."""
    rng = np.random.default_rng(1)
    for trial in range(3000):
        pred = rng.integers(0, 256, (8, 8)).astype(np.uint8)
        amp = int(rng.choice([40, 400, 3000]))
        full = rng.integers(-amp, amp + 1, 64).astype(np.int32)
        for variant, keep in ((1, [0]), (3, [0, 1, 8]), (16, [8 * k + m for k in range(4) for m in range(4)])):
            c = np.zeros(64, np.int32)
            c[keep] = full[keep]
            r1, a = _idct(8, c, variant, pred)
            r2, b = _idct(8, c, 64, pred)
            assert r1 == r2
            if r1 == 0:
                assert np.array_equal(a, b), (trial, variant)
        pred4 = pred[:4, :4]
        c4 = np.zeros(16, np.int32)
        c4[0] = full[0]
        r1, a = _idct(4, c4, 1, pred4)
        r2, b = _idct(4, c4, 16, pred4)
        assert r1 == r2 and (r1 != 0 or np.array_equal(a, b))


def test_dc_only_block_adds_rounded_dc():
    """H.264 identity: a DC-only block adds (dc+32)>>6 to every pixel (MD.cs:3712, :3789)."""
    for dc in (-4000, -65, -33, -32, -1, 0, 31, 32, 95, 640, 4100):
        pred = np.full((8, 8), 100, np.uint8)
        c = np.zeros(64, np.int32)
        c[0] = dc
        rc, out = _idct(8, c, 64, pred)
        exp = 100 + ((dc + 32) >> 6)
        if -64 <= exp - 0 and 0x40 + exp < 384 and 0x40 + exp >= 0:
            assert rc == 0 and np.all(out == min(max(exp, 0), 255))
        else:
            assert rc == -1  # clamp-table domain: the reference throws


def test_clamp_table_is_a_clamp():
    mm = T["mobi_vx2minmaxtable"]
    assert mm.tolist() == [0] * 64 + list(range(256)) + [255] * 64  # MobiConst.cs:587-621


def test_zigzag_tables_are_permutations_into_the_block():
    assert sorted(T["mobi_zz8"].tolist()) == list(range(64))
    assert sorted(T["mobi_zz4"].tolist()) == list(range(16))
    # scan positions 0, 0..2, 0..9 stay inside the regions the reduced IDCTs read (SURVEY hard part 5)
    assert T["mobi_zz8"][0] == 0
    assert set(T["mobi_zz8"][:3].tolist()) <= {0, 1, 8}
    assert all((i & 7) < 4 and (i >> 3) < 4 for i in T["mobi_zz8"][:10].tolist())
    assert T["mobi_zz4"][0] == 0


def test_cbp_maps_are_bijections():
    assert sorted(T["mobi_cbp_inter"].tolist()) == list(range(64))   # MD.cs:1809 <-> ME.cs:149
    assert sorted(T["mobi_cbp_intra"].tolist()) == list(range(64))   # MD.cs:1748 <-> ME.cs:407
    assert sorted(T["mobi_cbp4_inter"][1:].tolist()) == list(range(1, 16))  # ue>=1 (leading 0 bit), MD.cs:2904
    assert set(T["mobi_cbp4_intra"][1:17].tolist()) == set(range(16))       # MD.cs:2863


def test_dequant_tables_are_h264():
    """4x4: tbl = {10,13,16 / 11,14,18 / 13,16,20 / 14,18,23 / 16,20,25 / 18,23,29} by position class (H.264 8.5.9)."""
    base = {0: (10, 16, 13), 1: (11, 18, 14), 2: (13, 20, 16), 3: (14, 23, 18), 4: (16, 25, 20), 5: (18, 29, 23)}
    dq4, zz4 = T["mobi_dq4"].reshape(6, 16), T["mobi_zz4"]
    for m in range(6):
        for pos in range(16):
            idx = int(zz4[pos])
            r, c = idx >> 2, idx & 3
            cls = 0 if (r % 2 == 0 and c % 2 == 0) else (1 if (r % 2 == 1 and c % 2 == 1) else 2)
            assert dq4[m, pos] == base[m][cls], (m, pos)
    assert T["mobi_qdiv6"].tolist() == [q // 6 for q in range(54)]
    assert T["mobi_qmod6"].tolist() == [q % 6 for q in range(54)]


def test_residual_vlc_tables_are_prefix_consistent():
    """Every 12-bit LUT index that shares a code's (nbits-1)+sign prefix must hold the same entry
    (MobiConst.cs:10-14 format: E SSSSSS VVVVV BBBB) -- otherwise inverting the LUT would be ambiguous."""
    for name in ("mobi_vx2table0_a", "mobi_vx2table1_a"):
        A = T[name]
        seen = {}
        for i in range(4096):
            if (i >> 5) == 3:
                continue  # escape prefix 0000011 (MD.cs:3342)
            e = int(A[i])
            nb = e & 0xF
            if nb < 2 or ((e >> 4) & 0x1F) == 0:
                continue
            key = i >> (12 - (nb - 1))
            assert seen.setdefault((nb, key), e) == e, (name, i)
        # and the code set is prefix-free
        codes = sorted((nb - 1, key) for (nb, key) in seen)
        for (la, ka) in codes:
            for (lb, kb) in codes:
                if la < lb:
                    assert (kb >> (lb - la)) != ka, (name, la, ka, lb, kb)


def test_partition_luts_are_prefix_codes():
    lut, bits, shift, nlen = T["mobi_part_lut"], T["mobi_part_bits"], T["mobi_part_shift"], T["mobi_part_nbits_len"]
    for ver in range(2):
        for s in range(16):
            peek = 32 - int(shift[ver, s])
            for i in range(1 << peek):
                c = int(lut[ver, s, i])
                if c >= nlen[ver, s]:
                    continue
                nb = int(bits[ver, s, c])
                if nb == 0:
                    continue
                lo = (i >> (peek - nb)) << (peek - nb)
                assert all(int(lut[ver, s, j]) == c for j in range(lo, lo + (1 << (peek - nb)))), (ver, s, i)
            w, h = 16 >> (s // 4), 16 >> (s % 4)
            # shapes that cannot split further never decode a legal split code
            codes = {int(lut[ver, s, i]) for i in range(1 << peek)}
            legal = {c for c in codes if c < nlen[ver, s] and bits[ver, s, c] > 0}
            if h == 2:
                assert 8 not in legal
            if w == 2:
                assert 9 not in legal
            if s != 0:
                assert 6 not in legal and 7 not in legal


def test_copyblock_against_numpy_restatement():
    """CopyBlock (MD.cs:418-456): truncating half-pel, linear addressing, exceptions at the array bounds."""
    L = oracle_lib()
    rng = np.random.default_rng(3)
    S, H = 64, 40
    src = rng.integers(0, 256, S * H).astype(np.uint8)
    for _ in range(2000):
        w, h = int(rng.choice([16, 8, 4, 2, 1])), int(rng.choice([16, 8, 4, 2, 1]))
        off = int(rng.integers(0, S * (H - 16)))
        dx, dy = int(rng.integers(-40, 41)), int(rng.integers(-40, 41))
        dst = np.zeros(S * H, np.uint8)
        rc = L.mobi_oracle_copyblock(src.ctypes.data, src.size, dx, dy, w, h, dst.ctypes.data, dst.size, off, S)
        pos = off + (dy >> 1) * S + (dx >> 1)
        ph = (dx & 1) | ((dy & 1) << 1)
        last = pos + (h - 1) * S
        hi = last + w - 1 + (1 if ph & 1 else 0) + (S if ph & 2 else 0)
        bad = pos < 0 or hi >= src.size
        assert (rc != 0) == bad, (w, h, off, dx, dy, rc)
        if not bad:
            s32 = src.astype(np.int32)
            for i in range(h):
                p = pos + i * S
                g = lambda q: s32[q:q + w]
                if ph == 0:
                    exp = g(p)
                elif ph == 1:
                    exp = (g(p) >> 1) + (g(p + 1) >> 1)
                elif ph == 2:
                    exp = (g(p) >> 1) + (g(p + S) >> 1)
                else:
                    exp = (((g(p) >> 1) + (g(p + 1) >> 1)) >> 1) + (((g(p + S) >> 1) + (g(p + S + 1) >> 1)) >> 1)
                assert np.array_equal(dst[off + i * S: off + i * S + w], exp.astype(np.uint8))

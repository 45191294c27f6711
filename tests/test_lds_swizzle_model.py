"""Bank arithmetic of the attention kernels' LDS images, checked on the host.

The swizzles of ``csrc/lsh_attn_bwd.hip`` (``ab_sw`` / ``ab_off`` / ``ab_ds_off``; ``lsh_attn_fwd.hip`` and ``xattn.hip`` use the
same functions) are restated here and run through the LDS banking model of the MI355X notes: an access is served in
fixed lane groups, one LDS cycle per group when no two lanes of a group hit the same bank at different addresses;
64 banks of 4 bytes for ``ds_read_b128`` / ``ds_read_b64`` / ``ds_read_b64_tr_b16``, 32 banks for the stores.  This pins
the claims of DESIGN.md 5a (conflict-free fragment reads AND transposed reads of one image); it does not execute HIP."""
import itertools

import pytest


def ab_sw(row):
    return ((row >> 1) & 3) | ((((row >> 3) ^ (row >> 1)) & 1) << 2)


def ab_off(row, piece):
    return row * 128 + ((piece ^ ab_sw(row)) << 4)


def ab_ds_off(bs, key, gran):
    k0, k1, k2, k3 = key & 1, (key >> 1) & 1, (key >> 2) & 1, (key >> 3) & 1
    if bs == 128:
        return key * 256 + ((gran ^ ((k1 << 4) | (k0 << 3) | (k1 << 2) | (k2 << 1) | k3)) << 3)
    return key * 128 + ((gran ^ ((k1 << 3) | (k0 << 2) | (k2 << 1) | k3)) << 3)


def conflict_free(accesses, nbytes, banks):
    """accesses: byte addresses of the lanes of ONE lane group; every lane touches nbytes from its address."""
    owner = {}
    for a in accesses:
        for w in range(a // 4, (a + nbytes) // 4):
            if owner.setdefault(w % banks, w) != w:
                return False
    return True


B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
               [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]


@pytest.mark.parametrize("block,ks,hh", itertools.product(range(8), range(4), range(2)))
def test_fragment_reads_b128(block, ks, hh):
    """MFMA A/B fragments: lane (r, hh) reads piece ks*2+hh of row 32*block + r (ds_read_b128, 4 groups of 16 lanes)."""
    for grp in B128_GROUPS:
        assert conflict_free([ab_off(32 * block + r, ks * 2 + hh) for r in grp], 16, 64)


@pytest.mark.parametrize("base,dt,second", itertools.product(range(0, 256, 16), range(2), range(2)))
def test_transposed_reads_of_the_row_images(base, dt, second):
    """ds_read_b64_tr_b16 (2 groups of 32 lanes): lane -> row base + 4*hh + trq (+8 for the second read), 8-byte granule
    dt*8 + 4*trc + trp.  The kernel derives the second read's offset as tro[dt ^ 1] + 8 rows: checked against ab_off."""
    for hh in range(2):
        addrs = []
        for trq, trc, trp in itertools.product(range(4), range(2), range(4)):
            row = base + 4 * hh + trq + 8 * second
            piece, half = dt * 4 + 2 * trc + (trp >> 1), trp & 1
            addrs.append(ab_off(row, piece) + 8 * half)
            if second:    # the shortcut used in the kernels
                rl = 4 * hh + trq
                assert ab_off(row, piece) == base * 128 + 8 * 128 + ab_off(rl, (dt ^ 1) * 4 + 2 * trc + (trp >> 1))
        assert conflict_free(addrs, 8, 64)


@pytest.mark.parametrize("bs,kb,qt,second", [(bs, kb, qt, s) for bs in (64, 128) for kb in range(0, 2 * bs, 16)
                                             for qt in range(bs // 32) for s in range(2)])
def test_transposed_reads_of_the_ds_image(bs, kb, qt, second):
    """dQ phase: lane -> key kb + 8*hh + trq (+4), granule qt*8 + 4*trc + trp of the dS^T image."""
    for hh in range(2):
        addrs = [ab_ds_off(bs, kb + 8 * hh + trq + 4 * second, qt * 8 + 4 * trc + trp)
                 for trq, trc, trp in itertools.product(range(4), range(2), range(4))]
        assert conflict_free(addrs, 8, 64)


@pytest.mark.parametrize("bs,tile,qt,g,hh", [(bs, t, qt, g, hh) for bs in (64, 128) for t in range(2 * bs // 32)
                                             for qt in range(bs // 32) for g in range(4) for hh in range(2)])
def test_ds_image_stores(bs, tile, qt, g, hh):
    """Tile loop: lane (r, hh) stores 8 bytes at granule qt*8 + 2g + hh of key row 32*tile + r (ds_write_b64: groups of 16
    contiguous lanes, 32 banks); the kernel forms the offset as dso[g] ^ (qt << 6)."""
    for r0 in (0, 16):
        addrs = []
        for r in range(r0, r0 + 16):
            key = 32 * tile + r
            assert ab_ds_off(bs, key, qt * 8 + 2 * g + hh) == ab_ds_off(bs, key, 2 * g + hh) ^ (qt << 6)
            addrs.append(ab_ds_off(bs, key, qt * 8 + 2 * g + hh))
        assert conflict_free(addrs, 8, 32)


def test_padded_layouts_were_conflicted():
    """What the swizzles replaced: 144-byte rows are 2-way conflicted for the transposed reads, 272-byte dS^T rows 4-way."""
    rows144 = [(trq) * 144 + (4 * trc + trp) * 8 for trq, trc, trp in itertools.product(range(4), range(2), range(4))]
    rows272 = [(trq) * 272 + (4 * trc + trp) * 8 for trq, trc, trp in itertools.product(range(4), range(2), range(4))]
    assert not conflict_free(rows144, 8, 64) and not conflict_free(rows272, 8, 64)


@pytest.mark.parametrize("ks,tcol,second", itertools.product(range(2), range(0, 256, 32), range(2)))
def test_gemm_tn_ring_stage_reads(ks, tcol, second):
    """csrc/gemm_tn.hip, ring kernel: 512-byte rows, 64-byte chunks XOR-swizzled with (row & 3) on the DMA's source side;
    lane -> row 16*ks + 8*hh + trq (+4), byte column (tcol + 16*trc + 4*trp) * 2."""
    for hh in range(2):
        addrs = []
        for trq, trc, trp in itertools.product(range(4), range(2), range(4)):
            row = 16 * ks + 8 * hh + trq + 4 * second
            cb = (tcol + 16 * trc + 4 * trp) * 2
            addrs.append(row * 512 + ((((cb >> 6) ^ (row & 3)) << 6) | (cb & 63)))
        assert conflict_free(addrs, 8, 64)


def test_gemm_tn_ring_dma_fill_is_a_permutation_of_the_row():
    """A DMA instruction writes 2 rows in lane order: lane l lands on physical 16-byte piece l & 31 of row l >> 5 and fetches
    logical chunk ((l & 31) >> 2) ^ (row & 3); every logical piece of a row must be fetched exactly once."""
    for rowbase in range(0, 32, 2):
        for sub in range(2):
            row = rowbase + sub
            logical = sorted((((pp >> 2) ^ (row & 3)) << 2) | (pp & 3) for pp in range(32))
            assert logical == list(range(32))


# ---- the row staging of the backward kernel's epilogues (ab_stg_w / ab_stg_r of csrc/lsh_attn_bwd.hip)
def ab_stg_w(r, piece, hh):
    return r * 128 + ((piece ^ (r & 7)) << 4) + ((hh ^ ((r >> 3) & 1)) << 3)


def ab_stg_r(i, srow, spiece):
    return (i * 8 + srow) * 128 + ((spiece ^ srow) << 4)


B128_GROUPS_ALL = B128_GROUPS + [[l + 32 for l in g] for g in B128_GROUPS]


@pytest.mark.parametrize("piece", range(8))
def test_row_staging_stores_b64(piece):
    """lane (r, hh) stores 8 bytes of row r: ds_write_b64, four groups of 16 CONSECUTIVE lanes, 32 banks."""
    for first in range(0, 64, 16):
        acc = [ab_stg_w(lane & 31, piece, lane >> 5) for lane in range(first, first + 16)]
        assert conflict_free(acc, 8, 32)


@pytest.mark.parametrize("i", range(4))
def test_row_staging_reads_b128(i):
    """lane (srow = lane >> 3, spiece = lane & 7) reads a 16-byte piece of row 8 i + srow: ds_read_b128, four 16-lane groups, 64 banks."""
    for grp in B128_GROUPS_ALL:
        acc = [ab_stg_r(i, lane >> 3, lane & 7) for lane in grp]
        assert conflict_free(acc, 16, 64)


def test_row_staging_round_trip():
    """What the reader hands to the global store is the row in its natural order: piece q of row (8 i + srow), halves swapped back
    where bit 3 of the row (= i & 1) is set."""
    mem = {}
    for lane in range(64):
        r, hh = lane & 31, lane >> 5
        for piece in range(8):
            mem[ab_stg_w(r, piece, hh)] = (r, piece, hh)          # the 8-byte unit (row, piece, half)
    for i in range(4):
        for lane in range(64):
            srow, spiece = lane >> 3, lane & 7
            a = ab_stg_r(i, srow, spiece)
            lo, hi = mem[a], mem[a + 8]
            if i & 1:
                lo, hi = hi, lo
            assert lo == (8 * i + srow, spiece, 0) and hi == (8 * i + srow, spiece, 1)


def test_the_padded_staging_of_rounds_1_to_3_was_conflicted():
    acc = [(lane & 31) * 144 + (lane >> 5) * 8 for lane in range(16)]
    assert not conflict_free(acc, 8, 32)

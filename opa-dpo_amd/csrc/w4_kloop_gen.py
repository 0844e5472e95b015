#!/usr/bin/env python3
"""Generator of the K-loop of gemm_nt_w4_kernel (256x256x64 tile, four waves, bf16) as ONE inline-asm block with explicit registers.

    python opa-dpo_amd/csrc/w4_kloop_gen.py        ->  opa-dpo_amd/csrc/w4_kloop.inc   (committed; tests/test_abi_cpu.py checks it is current)

Why text and not HIP (round 5, profiles/r05_kloop_bisect.txt): hipcc's code for the source-level schedule of rounds 1-4 needed 562 quad-cycles per
K-tile and wave where the 128 MFMAs alone take 522; the same work as a hand-placed stream needs 531.  What the bisect against the vendor library's kernel of
this geometry named: (1) the CU's vector-memory path takes ONE 1-KiB LDS-DMA piece per wave every 64 cycles (4 waves x 1 KiB at 64 B/clk) - 13 pieces at a
period of 4 MFMAs are free, at a period of 3 they cost +17 quad-cycles, at 2 +66; (2) the landing wait + barrier of the next tile belong at MFMA 92, not 100,
so that the next tile's 16 fragment reads spread over 30 MFMA gaps with never two memory instructions in one gap; (3) everything the compiler adds around
asm statements (re-waits it cannot prove redundant, address arithmetic, a taken branch in mid-tile, accumulators out of order) costs another 14.

Register plan (physical; the C++ side binds its values with "{reg}" constraints - see W4K_* macros in the generated file):
  a[0:255]    accumulators: tile (i, j) = a[4 (8 i + j) : +3], i = A fragment (16 rows of the wave's 128), j = B fragment
  v[4:35]     B fragments (X, the inner MFMA index) of k-half 0, v[36:67] A fragments (Y) of k-half 0, v[68:99] / v[100:131] k-half 1
  v132 / v133 LDS read base of the B fragments, k-half 0 / 1;  v134 / v135 the same for A   (toggled between the stages by XOR 0x10000)
  v[140:147]  byte offsets of the 8 B pieces of this wave, v[148:155] of the 8 A pieces (per lane; row * ld + swizzled chunk)
  s[40:43]    buffer descriptor of B, s[44:47] of A;  s48 stage bit of the DMA target, s49 K byte offset of the tile being fetched
  s[52:59]    LDS byte offsets of the B pieces (low 16 bits of M0), s[60:67] of the A pieces
  s39 / s38   iterations of the steady-state loop with the first / second operand pair (the K-concatenated LoRA tail), s37 = (K-tiles >= 2)
The block expects the pieces of tiles 0 and 1 issued (C++ prologue; tile 1 only if there is one), waits for tile 0 and leaves the finished accumulators.
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# experiment variants (tools/build_kloop_exp.sh: W4K_EXP=<names> W4K_OUT=<file>; never the committed file).  noadv: the K offset never advances - every K-tile
# re-reads the first one (all L2 hits at the same operand randomness, results wrong: what memory stalls cost); early: the vendor kernel's skeleton for EVERY text
# (shipped for the deep-K products only, below); p3 / p2: the 13 pieces behind the stage release at a period of 3 / 2 MFMAs
EXP = [e for e in os.environ.get("W4K_EXP", "").split(",") if e]
EXP_KW = dict(early_release="early" in EXP, period=3 if "p3" in EXP else 2 if "p2" in EXP else 4)
for _e in EXP:                                             # eb<count>p<period>l<late>w<wait>: the vendor skeleton with another second burst / landing wait, for EVERY text
    import re as _re
    _m = _re.fullmatch(r"eb(\d+)p(\d+)l(\d+)w(\d+)", _e)
    if _m:
        EXP_KW.update(early_release=True, e_b2=(int(_m.group(1)), int(_m.group(2))), e_late=int(_m.group(3)), wait=int(_m.group(4)))
    _m = _re.fullmatch(r"w(\d+)", _e)
    if _m:
        EXP_KW.update(wait=int(_m.group(1)))
XOFF = [0, 1024, 256, 1280, 512, 1536, 768, 1792]      # LDS byte offset of B fragment j (B rows interleaved so that a lane owns 8 consecutive columns)


def frag_reg(op, kk, f):
    base = {("X", 0): 4, ("Y", 0): 36, ("X", 1): 68, ("Y", 1): 100}[(op, kk)]
    return f"v[{base + 4 * f}:{base + 4 * f + 3}]"


def mfma_ij(k):
    """(k-half, A fragment i, B fragment j) of the k-th MFMA of a tile.  Rows of 8 MFMAs share their A fragment; the B fragments are walked BACK AND FORTH (round 6), so
    the MFMA at a row change keeps its B operand too: +1.1 % on a bare power-limited MFMA loop (tools/micro/mfma_power.hip modes 0 / 2), +0.3-0.6 % on every GEMM shape
    (profiles/r06zz_ab_serp.txt).  Every accumulator still sees k-half 0 before k-half 1: bit-identical to the straight walk ("straight" experiment build)."""
    kk, i, j = k // 64, (k % 64) // 8, k % 8
    if (i & 1) and "straight" not in EXP:
        j = 7 - j
    return kk, i, j


def last_use_khalf0(op, f):
    """index of the last MFMA of a tile that reads k-half-0 fragment f of operand op"""
    return max(k for k in range(64) if (mfma_ij(k)[2] if op == "X" else mfma_ij(k)[1]) == f)


def mfma(k, zero_c=False):
    kk, i, j = mfma_ij(k)
    a = 4 * (8 * i + j)
    c = "0" if zero_c else f"a[{a}:{a + 3}]"
    return f"v_mfma_f32_16x16x32_bf16 a[{a}:{a + 3}], {frag_reg('Y', kk, i)}, {frag_reg('X', kk, j)}, {c}"


def slots_r5(first="X", spread_end=True, early_release=False, period=4, e_b2=(5, 3), e_late=3, wait=92):
    """slots[k] = what is issued after k MFMAs of the tile (slots[0]: before the first).  Items: ('rd', op, k-half, fragment) - k-half 1 of THIS tile,
    k-half 0 of the NEXT -, ('m0', n) / ('dma', n) for the n-th issued piece of tile t + 2, ('lgkm', n), ('vm', n), ('bar',), ('tog_rd',) read bases to
    the other stage, ('tog_m0',) DMA target to the other stage, ('salu',) the K advance.
    Default = the shipped schedule: two barriers, 13 pieces at a period of 4 MFMAs behind the stage release, landing wait at MFMA 92.
    early_release: the vendor kernel's skeleton instead (a third barrier releases the first operand's region after ITS reads; bursts of 5 pieces at period 3)."""
    F, G = (first, "Y" if first == "X" else "X")
    s = [[] for _ in range(129)]
    for n in range(8):
        # (B fragments in the order the previous tile's last row of MFMAs released their registers: that row walks j = 7 .. 0, see mfma_ij)
        s[1 + 2 * n].append(("rd", F, 1, 7 - n if F == "X" and "straight" not in EXP else n))
    s[2].append(("salu",))
    if early_release:
        s[16].append(("m0", 0))
        s[21].append(("lgkm", 0)); s[22].append(("bar",))
        for n in range(5):
            s[23 + 3 * n].append(("dma", n)); s[24 + 3 * n].append(("m0", n + 1)); s[25 + 3 * n].append(("rd", G, 1, n))
        s[39].append(("rd", G, 1, 5)); s[41].append(("rd", G, 1, 6)); s[43].append(("rd", G, 1, 7))
        s[51].append(("lgkm", 0)); s[52].append(("bar",))
        nb, pb = e_b2                                        # second burst: nb pieces at a period of pb MFMAs from 53; then e_late pieces every other MFMA in front of the wait
        for n in range(5, 5 + nb):
            s[53 + pb * (n - 5)].append(("dma", n)); s[54 + pb * (n - 5)].append(("m0", n + 1))
        n_before = 5 + nb + e_late
        for n in range(5 + nb, n_before):
            s[wait - 6 + 2 * (n - 5 - nb)].append(("dma", n))
            if n < 15:
                s[wait - 5 + 2 * (n - 5 - nb)].append(("m0", n + 1))
    else:
        for n in range(8):
            s[17 + 2 * n].append(("rd", G, 1, n))
        s[38].append(("m0", 0))
        s[39].append(("lgkm", 0)); s[40].append(("bar",))
        for n in range(13):
            s[41 + period * n].append(("dma", n)); s[42 + period * n].append(("m0", n + 1))
    if not early_release:
        n_before = 13
    s[min(85, wait - 7)].append(("tog_rd",))
    s[wait].append(("vm", n_before)); s[wait + 1].append(("bar",))
    if n_before == 16 or wait != 92:
        # no (or differently placed) pieces behind the wait: the 16 fragment reads of the next tile alone, then whatever pieces are left
        rd_slots = [wait + 2 + (7 * i) // 4 for i in range(16)]
        assert rd_slots[-1] <= 126, rd_slots
        order = {}
        for i, k in enumerate(rd_slots):
            order[k] = ("rd", "X" if i < 8 else "Y", 0, i % 8)
        free = [k for k in range(wait + 2, 127) if k not in order]
        left = list(range(n_before, 16))
        assert len(free) >= 2 * len(left)
        fi = 1
        for n in left:                                     # m0 of piece n was queued by the previous piece (or below for the first one behind the wait)
            order[free[fi]] = ("dma", n)
            if n < 15:
                order[free[fi + 1]] = ("m0", n + 1)
            fi += 3 if fi + 4 < len(free) else 2
        order[126 if 126 not in order else 125] = ("tog_m0",)
        s2 = sorted(order.items())
        for k, it in s2:
            s[k].append(it)
        s[127].append(("lgkm", 0))
        return s
    if spread_end:
        order = {94: ("rd", "X", 0, 0), 95: ("rd", "X", 0, 1), 97: ("rd", "X", 0, 2), 98: ("dma", 13), 99: ("rd", "X", 0, 3), 100: ("rd", "X", 0, 4), 101: ("m0", 14),
                 102: ("rd", "X", 0, 5), 103: ("dma", 14), 104: ("rd", "X", 0, 6), 105: ("rd", "X", 0, 7), 106: ("rd", "Y", 0, 0), 107: ("m0", 15), 108: ("rd", "Y", 0, 1),
                 109: ("dma", 15), 110: ("tog_m0",), 111: ("rd", "Y", 0, 2), 113: ("rd", "Y", 0, 3), 115: ("rd", "Y", 0, 4), 117: ("rd", "Y", 0, 5), 119: ("rd", "Y", 0, 6),
                 121: ("rd", "Y", 0, 7)}
    else:
        order = {94: ("rd", "X", 0, 0), 95: ("rd", "X", 0, 1), 96: ("rd", "X", 0, 2), 97: ("dma", 13), 98: ("rd", "X", 0, 3), 99: ("rd", "X", 0, 4), 100: ("m0", 14),
                 101: ("dma", 14), 102: ("m0", 15), 103: ("rd", "X", 0, 5), 104: ("rd", "X", 0, 6), 105: ("rd", "X", 0, 7), 106: ("rd", "Y", 0, 0), 107: ("rd", "Y", 0, 1),
                 110: ("rd", "Y", 0, 2), 113: ("rd", "Y", 0, 3), 115: ("rd", "Y", 0, 4), 118: ("rd", "Y", 0, 5), 121: ("rd", "Y", 0, 6), 124: ("rd", "Y", 0, 7), 125: ("dma", 15),
                 126: ("tog_m0",)}
    for k, it in order.items():
        s[k].append(it)
    s[127].append(("lgkm", 0))
    return s


def piece_order(first):
    return [(first, p) for p in range(8)] + [("Y" if first == "X" else "X", p) for p in range(8)]


def check_slots(slots, piece_ops):
    """the hazards the table must respect (LDS stage reuse, fragment registers, M0, vmcnt bookkeeping)"""
    flat = [(k, it) for k in range(129) for it in slots[k]]
    pos = {}
    for idx, (k, it) in enumerate(flat):
        pos.setdefault(it, []).append((k, idx))
    bars = [(k, idx) for idx, (k, it) in enumerate(flat) if it == ("bar",)]
    # every k-half-1 read precedes an lgkmcnt(0) + barrier that precedes the first piece into that operand's region
    for op in ("X", "Y"):
        last_rd = max(pos[("rd", op, 1, f)][0][1] for f in range(8))
        first_piece = min(pos[("dma", n)][0][1] for n in range(16) if piece_ops[n][0] == op)
        ok = False
        for (kb, ib) in bars:
            if last_rd < ib < first_piece and any(it == ("lgkm", 0) and last_rd < i2 < ib for i2, (k2, it) in enumerate(flat)):
                ok = True
        assert ok, f"region {op} not released before its first piece"
    # k-half 1 fragments: read before MFMA 64 with an lgkmcnt(0) in between; registers free (previous tile's MFMAs 64..127 are over)
    for op in ("X", "Y"):
        for f in range(8):
            k, idx = pos[("rd", op, 1, f)][0]
            assert k < 64 and any(it == ("lgkm", 0) and idx < i2 and k2 <= 64 for i2, (k2, it) in enumerate(flat)), (op, f)
    # next tile's k-half 0 fragments: behind the landing wait + barrier, behind tog_rd, register's last use (this tile's k-half 0) over, lgkmcnt(0) at the end
    ivm = next(i for i, (k, it) in enumerate(flat) if it[0] == "vm")
    ibar = next(i for i, (k, it) in enumerate(flat) if it == ("bar",) and i > ivm)
    itog = pos[("tog_rd",)][0][1]
    for op in ("X", "Y"):
        for f in range(8):
            k, idx = pos[("rd", op, 0, f)][0]
            last_use = last_use_khalf0(op, f)                      # MFMA index of the last k-half-0 use of that register
            assert idx > ibar and idx > itog and k > last_use, (op, f, k)
    assert flat[-1][1] == ("lgkm", 0) or any(it == ("lgkm", 0) for (k, it) in flat[-3:])
    # tog_rd after every k-half-1 read
    assert all(pos[("rd", op, 1, f)][0][1] < itog for op in ("X", "Y") for f in range(8))
    # pieces in order, each preceded by its own M0 write issued after the previous piece; exactly 13 pieces before the vmcnt(13)
    for n in range(16):
        im, idd = pos[("m0", n)][0][1], pos[("dma", n)][0][1]
        assert im < idd and (n == 0 or pos[("dma", n - 1)][0][1] < im), n
    assert sum(1 for i, (k, it) in enumerate(flat) if it[0] == "dma" and i < ivm) == flat[ivm][1][1]
    assert pos[("tog_m0",)][0][1] > pos[("dma", 15)][0][1]
    assert pos[("salu",)][0][1] < pos[("dma", 0)][0][1]


def rd(op, kk, f):
    base = {("X", 0): "v132", ("X", 1): "v133", ("Y", 0): "v134", ("Y", 1): "v135"}[(op, kk)]
    off = XOFF[f] if op == "X" else 2048 * f
    return f"ds_read_b128 {frag_reg(op, kk, f)}, {base}" + (f" offset:{off}" if off else "")


def tile_text(slots, piece_ops, mode):
    """mode 'loop': steady state; 'first': K-tile 0 of an output tile of the streaming kernel (a steady-state tile whose first MFMA on every accumulator
    starts from 0, outside the counted loop); 'tail_a': tile nt - 2 (no pieces, the landing wait is vmcnt(0)); 'tail_b': the last tile (no reads of a next tile)"""
    out = []
    first = mode == "first"
    if first:
        mode = "loop"
    for k in range(129):
        for it in slots[k]:
            if it[0] == "rd":
                if it[2] == 0 and mode == "tail_b":
                    continue
                out.append(rd(it[1], it[2], it[3]))
            elif it[0] == "dma":
                if mode == "loop":
                    op, p = piece_ops[it[1]]
                    out.append(f"buffer_load_dwordx4 v{(140 if op == 'X' else 148) + p}, {'s[40:43]' if op == 'X' else 's[44:47]'}, s49 offen lds")
            elif it[0] == "m0":
                if mode == "loop":
                    op, p = piece_ops[it[1]]
                    out.append(f"s_pack_ll_b32_b16 m0, s{(52 if op == 'X' else 60) + p}, s48")
            elif it[0] == "lgkm":
                out.append(f"s_waitcnt lgkmcnt({it[1]})")
            elif it[0] == "vm":
                if mode == "loop":
                    out.append(f"s_waitcnt vmcnt({it[1]})")
                elif mode == "tail_a":
                    out.append("s_waitcnt vmcnt(0)")
            elif it[0] == "bar":
                # the stage-release barriers only order the pieces behind the reads; the landing barrier is needed whenever a next tile is read
                is_landing = any(x[0] == "vm" for x in slots[k - 1])
                if mode == "loop" or (mode == "tail_a" and is_landing):
                    out.append("s_barrier")
            elif it[0] == "tog_rd":
                if mode != "tail_b":
                    out += [f"v_xor_b32_e32 v{r}, 0x10000, v{r}" for r in (132, 133, 134, 135)]
            elif it[0] == "tog_m0":
                if mode == "loop":
                    out.append("s_xor_b32 s48, s48, 1")
            elif it[0] == "salu":
                if mode == "loop":
                    out.append("s_add_u32 s49, s49, 0" if "noadv" in EXP else "s_add_u32 s49, s49, 0x80")
            else:
                raise ValueError(it)
        if k < 128:
            out.append(mfma(k, zero_c=first and k < 64))
            if mode == "loop" and not first and k == 125:
                out.append("s_sub_u32 s39, s39, 1")
            if mode == "loop" and not first and k == 126:
                out.append("s_cmp_lg_u32 s39, 0")
    if mode == "loop" and not first:
        out.append("s_cbranch_scc1 1b")
    return out


def block_text(first, early=False):
    slots = slots_r5(first=first, **dict(EXP_KW, early_release=early or EXP_KW["early_release"]))
    po = piece_order(first)
    check_slots(slots, po)
    t = []
    # tile 0 has landed for everyone (tiles 0 and 1 were issued by the C++ prologue; everything else the prologue computes sits in front of this wait)
    t += ["s_cmp_eq_u32 s37, 0", "s_cbranch_scc1 6f", "s_waitcnt vmcnt(16)", "s_branch 7f", "6:", "s_waitcnt vmcnt(0)", "7:", "s_barrier"]
    for op in ("X", "Y"):                                   # fragments of k-half 0 of tile 0
        for f in range(8):
            t.append(rd(op, 0, f))
    t += ["s_waitcnt lgkmcnt(0)", "s_cmp_eq_u32 s39, 0", "s_cbranch_scc1 3f", "1:"]
    t += tile_text(slots, po, "loop")
    t += ["3:", "s_cmp_eq_u32 s38, 0", "s_cbranch_scc1 4f"]
    # switch to the second operand pair (K-concatenated LoRA tail): addresses, descriptors, K origin; then the same loop again
    for p in range(16):
        t.append(f"v_mov_b32_e32 v{140 + p}, %[vo2_{p}]")
    for q in range(4):
        t.append(f"s_mov_b32 s{40 + q}, %[dx2_{q}]")
    for q in range(4):
        t.append(f"s_mov_b32 s{44 + q}, %[dy2_{q}]")
    t += ["s_mov_b32 s49, 0" if "noadv" in EXP else "s_mov_b32 s49, %[koff2]", "s_mov_b32 s39, s38", "s_mov_b32 s38, 0", "s_branch 1b", "4:", "s_cmp_eq_u32 s37, 0", "s_cbranch_scc1 5f"]
    t += tile_text(slots, po, "tail_a")
    t += ["5:"]
    t += tile_text(slots, po, "tail_b")
    t += ["s_nop 15", "s_nop 15", "s_nop 15", "s_nop 15"]     # the last MFMA results -> the epilogue's accumulator reads
    return t


def switch_text(tag):
    """addresses, descriptors and K origin of the operand pair fetched next (generic operands vo<tag>_p, dx<tag>_q, dy<tag>_q)"""
    t = [f"v_mov_b32_e32 v{140 + p}, %[vo{tag}_{p}]" for p in range(16)]
    t += [f"s_mov_b32 s{40 + q}, %[dx{tag}_{q}]" for q in range(4)] + [f"s_mov_b32 s{44 + q}, %[dy{tag}_{q}]" for q in range(4)]
    t += ["s_mov_b32 s49, 0" if "noadv" in EXP else "s_mov_b32 s49, -128"]                          # the loop adds 128 before its first piece: K-tile 0 of the new pair
    return t


def stream_text(first, early=False):
    """One OUTPUT tile of the streaming (persistent) kernel gemm_nt_w4s_kernel: the K-tile stream never drains at an output-tile boundary - the last two
    K-tiles of a tile fetch K-tiles 0 and 1 of the workgroup's NEXT output tile, its last K-tile reads the next tile's first fragments, then the (C++)
    epilogue stores this tile while those pieces are in flight.  K-tiles: FIRST (t = 0, accumulators start from 0) | s39 x LOOP from the first pair |
    switch, s38 x LOOP from the second pair (K-concatenated LoRA tail) | switch to the next tile's first pair, 2 x LOOP - or, on the workgroup's last tile
    (s37 = 0), the two draining tails.  s36 = 1 on the workgroup's first tile (wait for K-tile 0, read its fragments).  Needs K1 >= 3 K-tiles."""
    slots = slots_r5(first=first, **dict(EXP_KW, early_release=early or EXP_KW["early_release"]))
    po = piece_order(first)
    check_slots(slots, po)
    t = ["s_cmp_eq_u32 s36, 0", "s_cbranch_scc1 8f", "s_waitcnt vmcnt(16)", "s_barrier"]
    for op in ("X", "Y"):
        for f in range(8):
            t.append(rd(op, 0, f))
    t += ["s_waitcnt lgkmcnt(0)", "8:"]
    t += tile_text(slots, po, "first")
    t += ["s_mov_b32 s35, 0", "s_cmp_eq_u32 s39, 0", "s_cbranch_scc1 3f", "1:"]
    t += tile_text(slots, po, "loop")
    t += ["3:", "s_cmp_lg_u32 s35, 0", "s_cbranch_scc1 4f",
          "s_mov_b32 s35, 1", "s_cmp_eq_u32 s38, 0", "s_cbranch_scc1 4f"]
    t += switch_text("2") + ["s_mov_b32 s39, s38", "s_branch 1b"]
    t += ["4:", "s_cmp_lg_u32 s35, 1", "s_cbranch_scc1 9f",
          "s_mov_b32 s35, 2", "s_cmp_eq_u32 s37, 0", "s_cbranch_scc1 6f"]
    t += switch_text("3") + ["s_mov_b32 s39, 2", "s_branch 1b"]
    t += ["6:"]
    t += tile_text(slots, po, "tail_a")
    t += tile_text(slots, po, "tail_b")
    t += ["9:", "s_nop 15", "s_nop 15", "s_nop 15", "s_nop 15"]
    return t


def main():
    L = []
    L.append("// GENERATED by opa-dpo_amd/csrc/w4_kloop_gen.py - do not edit; re-run the generator after changing it.")
    L.append("// K-loop of gemm_nt_w4_kernel as one asm block (register plan, schedule and its hazard checks: see the generator).")
    for first, name in (("X", "W4K_TEXT_BFIRST"), ("Y", "W4K_TEXT_AFIRST")):
        txt = block_text(first)
        L.append(f"#define {name} \\")
        L += ['  "' + l + '\\n" \\' for l in txt]
        L.append('  ""')
    # the vendor library's skeleton (a third barrier releases the first operand's region after ITS reads, bursts of five pieces at a period of three MFMAs): ahead by
    # 1.5 - 2.2 % on the products of >= 128 K-tiles, level / behind on the shallow ones (profiles/r06v_ab_kloop_sched.txt) - the one-tile-per-workgroup kernel's text there
    for first, name in (("X", "W4K_TEXT_BFIRST_DEEP"), ("Y", "W4K_TEXT_AFIRST_DEEP")):
        txt = block_text(first, early=True)
        L.append(f"#define {name} \\")
        L += ['  "' + l + '\\n" \\' for l in txt]
        L.append('  ""')
    for first, name in (("X", "W4S_TEXT_BFIRST"), ("Y", "W4S_TEXT_AFIRST")):
        txt = stream_text(first)
        L.append(f"#define {name} \\")
        L += ['  "' + l + '\\n" \\' for l in txt]
        L.append('  ""')
    # ... and the streaming kernel's: with the default text streaming LOSES 4.7 % at >= 128 K-tiles, with this one it is 0.6 - 1 % ahead of one tile per workgroup
    for first, name in (("X", "W4S_TEXT_BFIRST_DEEP"), ("Y", "W4S_TEXT_AFIRST_DEEP")):
        txt = stream_text(first, early=True)
        L.append(f"#define {name} \\")
        L += ['  "' + l + '\\n" \\' for l in txt]
        L.append('  ""')
    outs = []
    for i in range(8):
        for j in range(8):
            n = 8 * i + j
            outs.append(f'"+{{a[{4 * n}:{4 * n + 3}]}}"(acc[{i}][{j}])')
    outs += [f'"+{{v{132 + q}}}"(w4k_rb[{q}])' for q in range(4)]
    outs += [f'"+{{v{140 + p}}}"(w4k_vo[{p}])' for p in range(16)]
    outs += [f'"+{{s{40 + q}}}"(w4k_dx[{q}])' for q in range(4)] + [f'"+{{s{44 + q}}}"(w4k_dy[{q}])' for q in range(4)]
    outs += ['"+{s48}"(w4k_stg)', '"+{s49}"(w4k_koff)', '"+{s39}"(w4k_na)', '"+{s38}"(w4k_nb)']
    ins = [f'"{{s{52 + p}}}"(w4k_pcx[{p}])' for p in range(8)] + [f'"{{s{60 + p}}}"(w4k_pcy[{p}])' for p in range(8)] + ['"{s37}"(w4k_has2)']
    ins += [f'[vo2_{p}] "v"(w4k_vo2[{p}])' for p in range(16)]
    ins += [f'[dx2_{q}] "s"(w4k_dx2[{q}])' for q in range(4)] + [f'[dy2_{q}] "s"(w4k_dy2[{q}])' for q in range(4)] + ['[koff2] "s"(w4k_koff2)']
    clob = [f'"v{i}"' for i in range(4, 132)] + ['"m0"', '"memory"', '"scc"']

    def wrap(name, items):
        L.append(f"#define {name} \\")
        for a in range(0, len(items), 6):
            L.append("  " + ", ".join(items[a:a + 6]) + ("," if a + 6 < len(items) else "") + " \\")
        L.append("")
    wrap("W4K_OUTS", outs)
    wrap("W4K_INS", ins)
    wrap("W4K_CLOBBERS", clob)
    L.append("#define W4K_RUN(TEXT) asm volatile(TEXT : W4K_OUTS : W4K_INS : W4K_CLOBBERS)")
    # streaming kernel: the fragments of k-half 0 (v[4:67]) live across the C++ epilogue between two output tiles -> in/out operands, not clobbers
    s_outs = list(outs) + [f'"+{{v[{4 + 4 * f}:{7 + 4 * f}]}}"(w4k_fr[{f}])' for f in range(16)]
    s_ins = [x for x in ins if "koff2" not in x] + ['"{s36}"(w4k_first)']
    s_ins += [f'[vo3_{p}] "v"(w4k_vo3[{p}])' for p in range(16)]
    s_ins += [f'[dx3_{q}] "s"(w4k_dx3[{q}])' for q in range(4)] + [f'[dy3_{q}] "s"(w4k_dy3[{q}])' for q in range(4)]
    s_clob = [f'"v{i}"' for i in range(68, 132)] + ['"s35"', '"m0"', '"memory"', '"scc"']
    wrap("W4S_OUTS", s_outs)
    wrap("W4S_INS", s_ins)
    wrap("W4S_CLOBBERS", s_clob)
    L.append("#define W4S_RUN(TEXT) asm volatile(TEXT : W4S_OUTS : W4S_INS : W4S_CLOBBERS)")
    out = os.environ.get("W4K_OUT") or os.path.join(HERE, "w4_kloop.inc")
    assert not EXP or os.environ.get("W4K_OUT"), "experiment variants go to W4K_OUT, not to the committed file"
    text = "\n".join(L) + "\n"
    if "--check" in os.sys.argv:
        assert open(out).read() == text, "w4_kloop.inc is stale: run python opa-dpo_amd/csrc/w4_kloop_gen.py"
        print("w4_kloop.inc is current")
        return
    open(out, "w").write(text)
    print("wrote", out, len(text), "bytes")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""K-loop emitter for the 256x256x64 / 4-wave bf16 GEMM tile (round 5).

One description of the K-loop of gemm_nt_w4_kernel as TEXT with explicit physical registers, built from three independent choices,
so that the bisect against hipBLASLt's kernel of the same geometry (tools/micro/kloop_bisect_gen.py) can swap ONE at a time:

  S  slot schedule: which non-MFMA instruction follows which of the 128 MFMAs of a K-tile ("vend": parsed from the vendor loop,
     "ours": parsed from hipcc's code for gemm_nt_w4_kernel, or a table given by the caller)
  L  LDS layout: "pad" (vendor: row-linear image, 16 B of padding per 8 rows, stage 0x10400) or "xor" (ours: 16-byte chunks XOR-swizzled
     by the row, B rows interleaved, stage 0x10000)
  D  DMA addressing: "sgpr" (one or a few address VGPRs per operand + one scalar offset per piece, descriptor base advanced per K-tile,
     M0 advanced by s_add_u32 after each piece) or "vgpr" (ours: 16 address VGPRs, one scalar K offset, M0 = s_pack_ll(piece offset, stage))

Register plan of the emitted text (the same for every variant):
  a[0:255]   accumulators, tile (i, j) = a[4 * (8 i + j) : +3]   (i = Y fragment = row block, j = X fragment = column block)
  v[4:35]    X fragments of k-half 0 (j = 0..7), v[36:67] Y fragments of k-half 0, v[68:99] X of k-half 1, v[100:131] Y of k-half 1
  v[132:139] fragment read bases (and the pad layout's toggle masks), v[140:155] DMA address VGPRs
  s[40:43] / s[44:47] buffer descriptors of X / Y, s48.. M0 bases / piece offsets / scalar offsets (see emit())
X = the operand whose fragments are the INNER index of the MFMA order (our B, the N side), Y = the outer one (our A, the M side).
"""
import re

XOFF_XOR = [0, 1024, 256, 1280, 512, 1536, 768, 1792]      # LDS byte offset of X fragment j in the "xor" layout (B rows interleaved)


def frag_reg(op, kk, f):
    base = {("X", 0): 4, ("Y", 0): 36, ("X", 1): 68, ("Y", 1): 100}[(op, kk)]
    return f"v[{base + 4 * f}:{base + 4 * f + 3}]"


def mfma(k):
    kk, i, j = k // 64, (k % 64) // 8, k % 8
    a = 4 * (8 * i + j)
    return f"v_mfma_f32_16x16x32_bf16 a[{a}:{a + 3}], {frag_reg('Y', kk, i)}, {frag_reg('X', kk, j)}, a[{a}:{a + 3}]"


# ---------------------------------------------------------------------------------------------------------------
# slot tables: list of 129 lists; slots[0] = before MFMA 0, slots[k + 1] = after MFMA k.  Items:
#   ('rd', op, kk, f)      fragment read (kk = 1: this tile, kk = 0: next tile)
#   ('m0', n)              M0 made ready for the n-th issued piece          ('dma', n)   the n-th issued piece
#   ('lgkm', n) ('vm', n) ('bar',)
#   ('tog_rd',)            read bases switch to the other stage             ('tog_m0',)  DMA target stage switches (after the tile's last piece)
#   ('salu',)              a slot where the original carries an addressing / loop scalar instruction
#   ('valu',)              ... a vector ALU instruction (address arithmetic)
# ---------------------------------------------------------------------------------------------------------------
def slots_from_vendor(vl):
    slots = [[] for _ in range(129)]
    k = 0; nd = 0
    body = vl[:-1]                                            # without the back edge
    for l in body:
        if l.startswith("v_mfma"):
            k += 1; continue
        m = re.match(r"ds_read_b128 v\[(\d+):\d+\], (v\d+)(?: offset:(\d+))?", l)
        if m:
            o = int(m.group(3) or 0)
            slots[k].append(("rd", "X" if m.group(2) == "v2" else "Y", (o % 128) // 64, o // 128)); continue
        if l.startswith("buffer_load_dwordx4"):
            slots[k].append(("dma", nd)); nd += 1; continue
        if l.startswith("s_mov_b32 m0") or l.startswith("s_add_u32 m0"):
            slots[k].append(("m0", nd)); continue             # ready for the NEXT piece (the one after the 16th is dropped by the emitter)
        if l.startswith("s_waitcnt lgkmcnt"):
            slots[k].append(("lgkm", int(re.search(r"\((\d+)\)", l).group(1)))); continue
        if l.startswith("s_waitcnt vmcnt"):
            slots[k].append(("vm", int(re.search(r"\((\d+)\)", l).group(1)))); continue
        if l.startswith("s_barrier"):
            slots[k].append(("bar",)); continue
        if l.startswith("v_xor_b32"):
            if not any(it == ("tog_rd",) for s in slots for it in s):
                slots[k].append(("tog_rd",))
            continue
        if l.startswith("s_xor_b32"):
            if not any(it == ("tog_m0",) for s in slots for it in s):
                pass
            slots[k].append(("tog_m0",) if "s47" in l else ("salu",)); continue
        if l.startswith("s_"):
            slots[k].append(("salu",)); continue
        raise ValueError(l)
    assert k == 128 and nd == 16
    return slots


def slots_from_ours(p19, p18, order_b=False):
    """p19: part of the tile from its first MFMA to the stage-release barrier (reads of k-half 1); p18: the rest"""
    slots = [[] for _ in range(129)]
    k = 0; nd = 0; nm = 0; nr = 0
    for part in (p19, p18):
        first_add = True
        for l in part:
            if l.startswith("v_mfma"):
                k += 1; continue
            if l.startswith("ds_read_b128"):
                kk = 1 if part is p19 else 0
                idx = nr % 16
                slots[k].append(("rd", "X" if idx < 8 else "Y", kk, idx % 8)); nr += 1; continue
            if l.startswith("buffer_load_dwordx4"):
                slots[k].append(("dma", nd)); nd += 1; continue
            if l.startswith("s_pack_ll_b32_b16 m0"):
                slots[k].append(("m0", nm)); nm += 1; continue
            if l.startswith("s_waitcnt lgkmcnt"):
                slots[k].append(("lgkm", int(re.search(r"\((\d+)\)", l).group(1)))); continue
            if l.startswith("s_waitcnt vmcnt"):
                slots[k].append(("vm", int(re.search(r"\((\d+)\)", l).group(1)))); continue
            if l.startswith("s_barrier"):
                slots[k].append(("bar",)); continue
            if l.startswith("v_add3_u32"):
                if part is p18 and first_add:
                    slots[k].append(("tog_rd",)); first_add = False
                else:
                    slots[k].append(("valu",))
                continue
            if l.startswith("s_cbranch") or l.startswith("s_branch"):
                continue
            if l.startswith("s_"):
                slots[k].append(("salu",)); continue
            raise ValueError(l)
    assert k == 128 and nd == 16 and nm == 16 and nr == 32, (k, nd, nm, nr)
    # the tile's last piece is followed by the stage switch of the DMA target
    last = max(i for i, s in enumerate(slots) if any(it[0] == "dma" for it in s))
    slots[last].append(("tog_m0",))
    return slots


def slots_r5(first="X", spread_end=True, early_release=True, period=3):
    """The round-5 schedule, written from the rules the bisect taught (profiles/r05_kloop_bisect.txt), not parsed from any kernel:
      * never more than ONE memory / M0 instruction in an MFMA gap;
      * the operand whose k-half-1 fragments are read first is released by its own barrier and takes the first DMA pieces, so the 13 pieces that
        must be in flight before the landing wait are spread over 70 MFMAs at a period of 3 gaps (piece, M0, fragment read);
      * the landing wait + barrier sit at MFMA 92, so the 16 fragment reads of the next tile spread over 30 gaps and the last one is 6 MFMAs old at the
        closing lgkmcnt(0).
    first: the operand (X = inner MFMA index, Y = outer) whose region is released first and whose pieces are issued first."""
    F, G = (first, "Y" if first == "X" else "X")
    s = [[] for _ in range(129)]
    for n in range(8):
        s[1 + 2 * n].append(("rd", F, 1, n))
    s[2].append(("salu",)); s[4].append(("salu",)); s[6].append(("salu",)); s[8].append(("salu",))
    nrel = 5 if early_release else 0
    if early_release:
        s[16].append(("m0", 0))
        s[21].append(("lgkm", 0)); s[22].append(("bar",))
        for n in range(5):
            s[23 + 3 * n].append(("dma", n)); s[24 + 3 * n].append(("m0", n + 1)); s[25 + 3 * n].append(("rd", G, 1, n))
        s[39].append(("rd", G, 1, 5)); s[41].append(("rd", G, 1, 6)); s[43].append(("rd", G, 1, 7))
        s[51].append(("lgkm", 0)); s[52].append(("bar",))
        for n in range(5, 10):
            s[53 + 3 * (n - 5)].append(("dma", n)); s[54 + 3 * (n - 5)].append(("m0", n + 1))
        for n in range(10, 13):
            s[86 + 2 * (n - 10)].append(("dma", n)); s[87 + 2 * (n - 10)].append(("m0", n + 1))
    else:
        for n in range(8):
            s[17 + 2 * n].append(("rd", G, 1, n))
        s[38].append(("m0", 0))
        s[39].append(("lgkm", 0)); s[40].append(("bar",))
        for n in range(13):
            s[41 + period * n].append(("dma", n)); s[42 + period * n].append(("m0", n + 1))
    s[85].append(("tog_rd",))
    s[92].append(("vm", 13)); s[93].append(("bar",))
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


# ---------------------------------------------------------------------------------------------------------------
# emitter
# ---------------------------------------------------------------------------------------------------------------
def emit(slots, L, D, piece_ops, extra_valu=True, loop_label="1", count_reg="s39"):
    """-> (prologue lines, loop lines).  piece_ops[n] = (op, p): the n-th issued piece of a tile is piece p (0..7) of operand op.
    Scalar registers: s[40:43] descX, s[44:47] descY, s39 loop count.
      D = sgpr: s48 / s49 M0 of piece 0 of X / Y in the target stage, s50 / s51 their stage toggle masks, s[52:58] / s[59:65] piece offsets 1..7 of X / Y
      D = vgpr: s48 stage bit of the target stage, s49 scalar K offset, s[52:59] / s[60:67] LDS piece offsets of X / Y
    Vector registers: L = pad: v132 / v133 read base X / Y, v134 / v135 toggle masks;  L = xor: v132 / v133 X k-half 0 / 1, v134 / v135 Y k-half 0 / 1
      D = sgpr: L = pad: v140 X, v141 Y address;  L = xor: v140 + (variant of the piece) ... see voff_of()
      D = vgpr: v[140:147] X pieces, v[148:155] Y pieces"""
    m0_step = 0x1040 if L == "pad" else 0x400

    def rd(op, kk, f):
        if L == "pad":
            base = "v132" if op == "X" else "v133"
            off = 128 * f + 64 * kk
        else:
            base = {("X", 0): "v132", ("X", 1): "v133", ("Y", 0): "v134", ("Y", 1): "v135"}[(op, kk)]
            off = XOFF_XOR[f] if op == "X" else 2048 * f
        return f"ds_read_b128 {frag_reg(op, kk, f)}, {base}" + (f" offset:{off}" if off else "")

    def voff_of(op, p):
        if D == "vgpr":
            return f"v{(140 if op == 'X' else 148) + p}"
        if L == "pad":
            return "v140" if op == "X" else "v141"
        # xor layout through scalar piece offsets: the source-side swizzle differs between pieces -> X: 4 address registers (p >> 1), Y: 2 (p & 1)
        return f"v{140 + (p >> 1)}" if op == "X" else f"v{144 + (p & 1)}"

    def dma(n):
        op, p = piece_ops[n]
        desc = "s[40:43]" if op == "X" else "s[44:47]"
        if D == "vgpr":
            return f"buffer_load_dwordx4 {voff_of(op, p)}, {desc}, s49 offen lds"
        so = "0" if p == 0 else f"s{(52 if op == 'X' else 59) + p - 1}"
        return f"buffer_load_dwordx4 {voff_of(op, p)}, {desc}, {so} offen lds"

    def m0(n):
        if n >= 16:
            return None
        op, p = piece_ops[n]
        if D == "vgpr":
            return f"s_pack_ll_b32_b16 m0, s{(52 if op == 'X' else 60) + p}, s48"
        if p == 0:
            return f"s_mov_b32 m0, {'s48' if op == 'X' else 's49'}"
        return f"s_add_u32 m0, m0, {m0_step:#x}"

    def tog_rd():
        if L == "pad":
            return ["v_xor_b32_e32 v132, v134, v132", "v_xor_b32_e32 v133, v135, v133"]
        return [f"v_xor_b32_e32 v{r}, 0x10000, v{r}" for r in (132, 133, 134, 135)]

    def tog_m0():
        if D == "vgpr":
            return ["s_xor_b32 s48, s48, 1"]
        return ["s_xor_b32 s48, s50, s48", "s_xor_b32 s49, s51, s49"]

    def advance():
        if D == "vgpr":
            return ["s_add_u32 s49, s49, 0x80"]
        return ["s_add_u32 s40, s40, 0x80", "s_addc_u32 s41, s41, 0", "s_add_u32 s44, s44, 0x80", "s_addc_u32 s45, s45, 0"]

    # ---- prologue: tiles 0 and 1 into stages 0 and 1, fragments of k-half 0 of tile 0
    pro = []
    for tile in (0, 1):
        for n in range(16):
            pro += [m0(n), dma(n)]
        pro += tog_m0()
        if tile == 0:
            pro += advance()          # the loop advances BEFORE its first piece: leave the K position at tile 1
    pro += ["s_waitcnt vmcnt(16)", "s_barrier"]
    for op in ("X", "Y"):
        for f in range(8):
            pro.append(rd(op, 0, f))
    pro += ["s_waitcnt lgkmcnt(0)"]
    # ---- loop
    pending_salu = advance()          # the tile's addressing arithmetic goes into the first 'salu' slots (before the first piece in both schedules)
    loop = []
    first_dma_seen = False

    def put(items):
        nonlocal first_dma_seen
        for it in items:
            if it[0] == "rd":
                loop.append(rd(it[1], it[2], it[3]))
            elif it[0] == "dma":
                if pending_salu:                       # must not happen after the first piece: flush before it
                    loop.extend(pending_salu); pending_salu.clear()
                loop.append(dma(it[1])); first_dma_seen = True
            elif it[0] == "m0":
                x = m0(it[1])
                if x:
                    loop.append(x)
            elif it[0] == "lgkm":
                loop.append(f"s_waitcnt lgkmcnt({it[1]})")
            elif it[0] == "vm":
                loop.append(f"s_waitcnt vmcnt({it[1]})")
            elif it[0] == "bar":
                loop.append("s_barrier")
            elif it[0] == "tog_rd":
                loop.extend(tog_rd())
            elif it[0] == "tog_m0":
                loop.extend(tog_m0())
            elif it[0] == "salu":
                if pending_salu:
                    loop.append(pending_salu.pop(0))
            elif it[0] == "valu":
                if extra_valu:
                    loop.append("v_mov_b32_e32 v156, v157")
            elif it[0] == "raw":
                loop.append(it[1])
            else:
                raise ValueError(it)
    for k in range(129):
        put(slots[k])
        if k < 128:
            loop.append(mfma(k))
            if k == 125:
                loop.append(f"s_sub_u32 {count_reg}, {count_reg}, 1")
            if k == 126:
                loop.append(f"s_cmp_lg_u32 {count_reg}, 0")
    assert not pending_salu
    loop.append(f"s_cbranch_scc1 {loop_label}b")
    return [x for x in pro if x], loop


CPP_SETUP = r"""
  const int lane = threadIdx.x & 63, wave = rfl(threadIdx.x >> 6);
  int tm, tn; tile_of(tm, tn);
  // K-start stagger experiment: the low 7 bits of `ld` = K-tiles by which consecutive column tiles are shifted (0 = everyone starts at k = 0)
  const int stag = ld & 127; ld &= ~127;
  const size_t koff0 = (size_t)((tn & 7) * stag) * 128;
  const u64 xb = (u64)(B + (size_t)tn * 256 * ld + koff0), yb = (u64)(A + (size_t)tm * 256 * ld + koff0);
  const unsigned xlo = rfl((int)xb), xhi = rfl((int)(xb >> 32)), ylo = rfl((int)yb), yhi = rfl((int)(yb >> 32));
  const int wr = wave >> 1, wc = wave & 1, srow = lane >> 3, spos = lane & 7, frow = lane & 15, fchk = lane >> 4, fsw = (frow >> 1) & 7;
"""


def inputs_for(L, D):
    """-> (C++ setup text, dict physical register -> C expression)"""
    cpp = CPP_SETUP
    ins = {"s39": "niter", "s40": "xlo", "s41": "xhi", "s42": "-1", "s43": "0x00020000", "s44": "ylo", "s45": "yhi", "s46": "-1", "s47": "0x00020000"}
    if L == "pad":
        cpp += """  const unsigned rx = wc * 128 + frow * 8, ry = wr * 128 + frow * 8;
  const unsigned rbx = rx * 128 + (rx >> 3) * 16 + fchk * 16, rby = 0x8200 + ry * 128 + (ry >> 3) * 16 + fchk * 16;
  const unsigned m0x = rfl(wave * 0x410), m0y = rfl(0x8200 + wave * 0x410), stage = 0x10400;
"""
        ins.update({"v132": "rbx", "v133": "rby", "v134": "(rbx ^ (rbx + 0x10400))", "v135": "(rby ^ (rby + 0x10400))"})
    else:
        cpp += """  const unsigned rowX = 0x8000 + (wc * 128 + (frow >> 1) * 16 + (frow & 1)) * 128, rowY = (wr * 128 + frow) * 128;
  const unsigned c0 = ((0 + fchk) ^ fsw) << 4, c1 = ((4 + fchk) ^ fsw) << 4;
  const unsigned m0x = rfl(0x8000 + wave * 8192), m0y = rfl(wave * 8192), stage = 0x10000;
"""
        ins.update({"v132": "(rowX + c0)", "v133": "(rowX + c1)", "v134": "(rowY + c0)", "v135": "(rowY + c1)"})
    # global rows of a piece.  pad layout: piece p of wave w = tile rows 32 p + 8 w + srow, plain chunks.  xor layout: ours (see gemm.hip set_voff)
    if L == "pad":
        cpp += """  unsigned vx[8], vy[8], sx[8], sy[8];
  for (int p = 0; p < 8; ++p) { sx[p] = sy[p] = (unsigned)(p * 32 * ld); vx[p] = vy[p] = (unsigned)((wave * 8 + srow) * ld + spos * 16); }
"""
    else:
        cpp += """  unsigned vx[8], vy[8], sx[8], sy[8];
  for (int p = 0; p < 8; ++p) {
    sy[p] = (unsigned)(p * 8 * ld); vy[p] = (unsigned)(wave * 64 + srow) * ld + (unsigned)((spos ^ (((p & 1) * 4 + (srow >> 1)) & 7)) * 16);
    sx[p] = (unsigned)(((p >> 1) * 16 + (p & 1)) * ld); vx[p] = (unsigned)(wave * 64 + (srow & 1) * 8 + (srow & 6)) * ld + (unsigned)((spos ^ ((wave * 4 + (p >> 1)) & 7)) * 16);
  }
"""
    if D == "sgpr":
        ins.update({"s48": "m0x", "s49": "m0y", "s50": "(m0x ^ (m0x + stage))", "s51": "(m0y ^ (m0y + stage))"})
        for p in range(1, 8):
            ins[f"s{52 + p - 1}"] = f"sx[{p}]"
            ins[f"s{59 + p - 1}"] = f"sy[{p}]"
        if L == "pad":
            ins.update({"v140": "vx[0]", "v141": "vy[0]"})
        else:
            for q in range(4):
                ins[f"v{140 + q}"] = f"vx[{2 * q}]"
            ins.update({"v144": "vy[0]", "v145": "vy[1]"})
    else:
        ins.update({"s48": "0", "s49": "0"})
        step = 0x1040 if L == "pad" else 0x400
        for p in range(8):
            # the pack form holds the stage in the HIGH half of M0: only the xor layout's 64-KiB stage fits it; with the pad layout the
            # "stage bit" selects + 0x10000 as well (timing harness only: 2 x 0x10400 would not fit otherwise) -> stage 1 starts at 0x10000 + piece
            ins[f"s{52 + p}"] = f"(unsigned)rfl((int)(m0x + {p} * {step}))"
            ins[f"s{60 + p}"] = f"(unsigned)rfl((int)(m0y + {p} * {step}))"
            ins[f"v{140 + p}"] = f"(vx[{p}] + sx[{p}])"
            ins[f"v{148 + p}"] = f"(vy[{p}] + sy[{p}])"
    return cpp, ins


# the rule-built schedule that SHIPS lives in the product generator: the harness always times that table
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))), "opa-dpo_amd", "csrc"))
from w4_kloop_gen import slots_r5, piece_order      # noqa: E402,F401,F811

#!/usr/bin/env python3
"""Generator of the hand-placed instruction streams of the d == 128 attention forward (metamorph_amd/csrc/attn4.hip).

One wave per SIMD owns 64 query rows (two 32-row blocks) and the whole 512-entry register file.  Per K / V tile of 64 keys a wave issues
64 `v_mfma_f32_32x32x16_bf16` (32 for S^T = K Q^T of tile t + 1, 32 for O^T += V^T P^T of tile t); everything else -- the exponentials /
packing / row sums of tile t, the row maxima of tile t + 1, the LDS fragment reads into the accumulator file and the LDS-DMA of the tiles
ahead -- is placed BETWEEN those MFMAs, at most ~5 single-issue instructions per 32-cycle MFMA slot (CDNA4 guide, "one wave per SIMD").
hipcc will not build that interleave (256 registers + 300 spills, or tuple copies on every loop back-edge), so the
stream is written as one `asm volatile` per instruction in program order ON LITERAL REGISTERS:

    a[0:127]    O^T accumulators, block (db, qb) at 16 * (db * 2 + qb)
    a[128:191]  Q fragments as stored (B operands of the score MFMAs), (qb, ks) at 128 + 4 * (qb * 8 + ks)
    a[192:223]  K fragment ring (8 fragments, (ks % 4) * 2 + kb), filled by ds_read_b128
    a[224:255]  V fragment ring (8 fragments, (kstep % 2) * 4 + db), filled by ds_read_b64_tr_b16 pairs
    v[64:191]   two score tiles S[parity][kb][qb] (16 registers each)
    v[192:223]  -m (running row maximum, in units of the RAW dot product q . k) as the C operand of a score chain, one 16-register tuple per qb
    v[224:255]  P^T fragments (bf16 pairs), (kstep, qb) at 224 + 4 * (kstep * 2 + qb)
    v[0:63]     everything hipcc allocates itself (the kernel carries amdgpu_num_vgpr(64); tools/audit_attn4.py checks the emitted
                code object for any compiler instruction that touches v64+ or an accumulator register)

    python tools/gen_attn4.py            # rewrites metamorph_amd/csrc/attn4_gen/*.inc

Emitted blocks (included inside attn4::fwd_kernel, which declares the few compiler-allocated values they name: KA, VA, LS, MX, lim2, ...):
    head         QK(0), row maxima + first maximum                                   (step -1)
    loop<p>      QK(t + 1) || softmax-finish(t)   then   PV(t) || row maxima(t + 1)     (step t, parity p = t & 1)
    tail<p>      softmax-finish(tw) then PV(tw)                                      (a wave's last tile)
    zero_o, q_load, q_pre01, epi_read   accumulator initialisation / Q rows -> v[128:191] / Q fragments of d-steps 0, 1 -> accumulator file / read-out
                 (the fragments of d-steps 2..7 are placed under the head's score MFMAs)

Arithmetic (round 5): the score chains run on the operands AS STORED -- s' = q . k - m is the fp32 product of the bf16 values the reference's
SDPA sees -- and the softmax scale enters on the fp32 side: P = exp2((scale * log2 e) * s'), one `v_mul_f32` per score one MFMA gap ahead
of its `v_exp_f32` (the constant sits in an SGPR).  Rounds 1-4 multiplied q by scale * log2 e and re-rounded it to bf16 before
the MFMA: one rounding more than the reference has, a score error that grows with |s| (measured: 3x the error of a textbook bf16 flash
attention at |s| = 50).  Costs 64 multiplies per tile.  They are PLAIN `v_mul_f32`, not `v_pk_mul_f32`: between MFMAs a
packed-fp32 instruction costs ~22 cycles here against ~3.5 for a plain one (measured with the timing ablations `--abl noscl / noadd /
sclscalar`: 32 v_pk_mul_f32 = +700 cycles per tile, 64 v_mul_f32 = +220); the same holds for `v_pk_add_f32`, so the row sums stay scalar.
`safe_*` are the same streams with every LDS read waited for at once and every MFMA followed by 32 wait states: the debugging build that
separates a placement / hazard defect from a logic defect (tools/bench_attn4.py runs both).
"""
import os
import sys

ABL = set()                                                  # timing-only ablations (--abl a,b,...): the build is WRONG by construction
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "metamorph_amd", "csrc", "attn4_gen")
TILE = 16384
O_A, Q_A, KRING_A, VRING_A = 0, 128, 192, 224
S_V, NM_V, P_V = 64, 192, 224


def areg(lo, n):
    return "a[%d:%d]" % (lo, lo + n - 1)


def vreg(lo, n):
    return "v[%d:%d]" % (lo, lo + n - 1)


def o_blk(db, qb):
    return O_A + 16 * (db * 2 + qb)


def q_frag(qb, ks):
    return Q_A + 4 * (qb * 8 + ks)


def k_frag(ks, kb):
    return KRING_A + 4 * ((ks % 4) * 2 + kb)


def v_frag(kstep, db):
    return VRING_A + 4 * ((kstep % 2) * 4 + db)


def s_blk(par, kb, qb):
    return S_V + 16 * (par * 4 + kb * 2 + qb)


def nm_blk(qb):
    return NM_V + 16 * qb


def p_frag(kstep, qb):
    return P_V + 4 * (kstep * 2 + qb)


class Stream:
    def __init__(self, safe):
        self.safe = safe
        self.lines = []
        self.issued = 0                                      # LDS instructions issued so far (ids 1..issued)
        self.done = 0                                        # ids <= done are known complete
        self.stats = []
        self.max_out = 0

    def raw(self, text):
        self.lines.append(text)

    def asm(self, text, outs="", ins="", clobbers=""):
        s = 'asm volatile("%s"' % text
        if outs or ins or clobbers:
            s += " : %s : %s" % (outs, ins)
            if clobbers:
                s += " : " + clobbers
        self.lines.append(s + ");")

    def lds(self, n=1):
        self.issued += n
        self.max_out = max(self.max_out, self.issued - self.done)
        if self.safe:
            self.asm("s_waitcnt lgkmcnt(0)")
            self.done = self.issued
        return self.issued

    def wait(self, ident):
        if ident <= self.done or "nolds" in ABL:
            return
        n = min(self.issued - ident, 15)
        self.asm("s_waitcnt lgkmcnt(%d)" % n)
        self.done = self.issued - n

    def mfma_pad(self):
        if self.safe:
            self.asm("s_nop 15\\n\\ts_nop 15")


def elem(e):
    """softmax-finish order = consumption order of the P fragments: fragment f = kstep * 2 + qb, element m = 0..7 -> (kstep, qb, kb, r)"""
    f, m = e >> 3, e & 7
    kstep, qb = f >> 1, f & 1
    kb, s = kstep >> 1, kstep & 1
    return kstep, qb, kb, 8 * s + m


def emit_mask(st, nxt):
    """S[nxt] element (kb, qb, r) is key kb*32 + 8*(r >> 2) + 4*hi + (r & 3): -inf where that exceeds lim2[qb] (hi folded into lim2).
    VALU write of VCC -> VALU read of VCC needs two wait states on gfx950 (hipcc pads nothing inside asm)."""
    st.raw("if (mask_next) {")
    for kb in range(2):
        for qb in range(2):
            for r in range(16):
                key = kb * 32 + 8 * (r >> 2) + (r & 3)
                v = s_blk(nxt, kb, qb) + r
                st.asm("v_cmp_gt_i32 vcc, %d, %%0\\n\\ts_nop 1\\n\\tv_cndmask_b32 v%d, v%d, %%1, vcc" % (key, v, v), "", '"v"(lim2[%d]), "v"(ninf)' % qb, '"vcc"')
    st.raw("}")


def emit_dec_step(st, i):
    """row maxima of tile t + 1 across the two lanes of a row, and the wave-wide "some row grew by more than 2^THR" mask: five small steps
    placed under the last P V MFMAs (a v_permlane32_swap needs its operands two wait states old; the branch reads an SGPR pair written long before)"""
    if i == 0:
        st.asm("v_max_f32 %0, %2, %3\\n\\tv_max_f32 %1, %4, %5", '"=&v"(rm_[0]), "=&v"(rm_[1])', '"v"(MX[0][0]), "v"(MX[0][1]), "v"(MX[1][0]), "v"(MX[1][1])')
    elif i == 1:
        st.asm("v_mov_b32 %0, %2\\n\\tv_mov_b32 %1, %3", '"=&v"(rt_[0]), "=&v"(rt_[1])', '"v"(rm_[0]), "v"(rm_[1])')
    elif i == 2:
        st.asm("v_permlane32_swap_b32 %0, %1\\n\\tv_permlane32_swap_b32 %2, %3", '"+v"(rm_[0]), "+v"(rt_[0]), "+v"(rm_[1]), "+v"(rt_[1])', "")
    elif i == 3:
        st.asm("v_max_f32 %0, %0, %2\\n\\tv_max_f32 %1, %1, %3", '"+v"(rm_[0]), "+v"(rm_[1])', '"v"(rt_[0]), "v"(rt_[1])')
    else:
        st.asm("v_max_f32 %1, %2, %3\\n\\ts_nop 0\\n\\tv_cmp_lt_f32 %0, %4, %1", '"=s"(grow_), "=&v"(rt_[0])', '"v"(rm_[0]), "v"(rm_[1]), "v"(thr_)')


def emit_decide(st, nxt, head):
    st.raw("{")
    if head:
        st.raw("  rm_[0] = half_swap_max(max2_raw(MX[0][0], MX[0][1])); rm_[1] = half_swap_max(max2_raw(MX[1][0], MX[1][1]));")
    if head:                                                 # first tile: the running maximum becomes the tile's row maximum (the chain ran with C = 0)
        st.raw("  rm_[0] = rm_[0] == -INFINITY ? 0.f : rm_[0]; rm_[1] = rm_[1] == -INFINITY ? 0.f : rm_[1];")
        st.raw("  {")
    else:                                                    # later tiles: s' is relative to the running maximum; move it only when a row grew by > 2^THR
        st.raw("  if (grow_ != 0) {")
        st.raw("    ATTN4_COUNT_RESCALE();")                # wave-uniform tally of this branch (scalar ALU): tests assert it fired (Args::dbg)
        st.raw("    rm_[0] = fmaxf(rm_[0], 0.f); rm_[1] = fmaxf(rm_[1], 0.f);")
        st.raw("    const float al_[2] = {__builtin_amdgcn_exp2f(-rm_[0] * sl2), __builtin_amdgcn_exp2f(-rm_[1] * sl2)};")   # rm_ is in raw q . k units
        st.asm("s_nop 15\\n\\ts_nop 15\\n\\ts_nop 15\\n\\ts_nop 15\\n\\ts_nop 15\\n\\ts_nop 15\\n\\ts_nop 15\\n\\ts_nop 15")    # the P V MFMAs drain
        st.raw("    float t0_, t1_;")
        for qb in range(2):
            for db in range(4):
                for r in range(0, 16, 2):
                    a0 = o_blk(db, qb) + r
                    st.asm("v_accvgpr_read_b32 %%0, a%d\\n\\tv_accvgpr_read_b32 %%1, a%d\\n\\tv_mul_f32 %%0, %%0, %%2\\n\\tv_mul_f32 %%1, %%1, %%2\\n\\t"
                           "v_accvgpr_write_b32 a%d, %%0\\n\\tv_accvgpr_write_b32 a%d, %%1" % (a0, a0 + 1, a0, a0 + 1),
                           '"=&v"(t0_), "=&v"(t1_)', '"v"(al_[%d])' % qb)
            st.raw("    LS[%d][0] *= al_[%d]; LS[%d][1] *= al_[%d]; LS[%d][2] *= al_[%d]; LS[%d][3] *= al_[%d];" % ((qb, qb) * 4))
    for qb in range(2):
        for kb in range(2):
            for r in range(16):
                v = s_blk(nxt, kb, qb) + r
                st.asm("v_sub_f32 v%d, v%d, %%0" % (v, v), "", '"v"(rm_[%d])' % qb)
        for r in range(16):
            v = nm_blk(qb) + r
            if head:
                st.asm("v_sub_f32 v%d, 0, %%0" % v, "", '"v"(rm_[%d])' % qb)
            else:
                st.asm("v_sub_f32 v%d, v%d, %%0" % (v, v), "", '"v"(rm_[%d])' % qb)
    st.asm("s_nop 7")                                        # VALU write of the -m tuples -> C operand of the next score chain
    st.raw("  }")
    st.raw("}")


def gen(mode, par, safe):
    """mode: 'head' | 'loop' | 'tail'"""
    qk = mode in ("loop", "head")
    sm = mode in ("loop", "tail")
    pv = mode in ("loop", "tail")
    mx = mode in ("loop", "head")
    head = mode == "head"
    nxt = par ^ 1
    st = Stream(safe)
    st.raw("// ---- generated by tools/gen_attn4.py: mode %s, parity %d%s ----" % (mode, par, ", SAFE (serialised)" if safe else ""))

    kread_id = {}
    kslot_cur = nxt * TILE                                   # K(t + 1) lives in ring slot (t + 1) & 1

    def kread(ks, kb, slot_off):
        r = k_frag(ks, kb)
        st.asm("ds_read_b128 %s, %%0 offset:%d" % (areg(r, 4), slot_off + kb * 8192), "", '"v"(KA[%d])' % ks)
        return st.lds()

    if head:                                                 # nobody pre-read K(0): do it here
        for ks in range(4):
            for kb in range(2):
                kread_id[(ks, kb)] = kread(ks, kb, kslot_cur)
    else:                                                    # issued by the previous step's phase B, in this order
        for ks in range(4):
            for kb in range(2):
                st.issued += 1
                kread_id[(ks, kb)] = st.issued
        if safe:
            st.done = st.issued

    vread_id = {}

    def vstmt(kstep, db):
        r = v_frag(kstep, db)
        off = par * TILE + kstep * 4096
        st.asm("ds_read_b64_tr_b16 %s, %%0 offset:%d\\n\\tds_read_b64_tr_b16 %s, %%0 offset:%d" % (areg(r, 2), off, areg(r + 2, 2), off + 2048),
               "", '"v"(VA[%d])' % db)
        vread_id[(kstep, db)] = st.lds(2)

    def do_filler(f):
        kind = f[0]
        if ("no" + kind) in ABL or (kind in ("kread", "vread", "kpre") and "nolds" in ABL):
            return
        if kind == "exp":
            _, kb, qb, r = f
            v = s_blk(par, kb, qb) + r
            st.asm("v_exp_f32 v%d, v%d" % (v, v))
        elif kind == "scl":                                  # s' (raw q . k units) -> log2 domain; the constant lives in an SGPR.  Two plain multiplies:
            _, kb, qb, r = f                                 # a v_pk_mul_f32 costs ~22 cycles between MFMAs (measured, profiles/r5_attn4_packed_fp32.log), these two ~7
            v = s_blk(par, kb, qb) + r
            st.asm("v_mul_f32 v%d, %%0, v%d\\n\\tv_mul_f32 v%d, %%0, v%d" % (v, v, v + 1, v + 1), "", '"s"(sl2b_)')
        elif kind == "cvt":
            pair = f[1]
            kstep, qb, kb, r = elem(2 * pair)
            v = s_blk(par, kb, qb) + r
            st.asm("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (p_frag(kstep, qb) + (pair & 3), v, v + 1))
        elif kind == "kread":
            _, ks, kb = f
            kread_id[(ks, kb)] = kread(ks, kb, kslot_cur)
        elif kind == "vread":
            vstmt(f[1], f[2])
        elif kind == "add":
            _, kb, qb, r = f
            st.asm("v_add_f32 %%0, %%0, v%d" % (s_blk(par, kb, qb) + r), '"+v"(LS[%d][%d])' % (qb, r & 3), "")    # four partial sums: no dependent chain
        elif kind == "max":
            _, qb, kb, step = f
            s = s_blk(nxt, kb, qb)
            if step == 0:
                st.asm("v_max3_f32 %%0, v%d, v%d, v%d" % (s, s + 1, s + 2), '"=v"(MX[%d][%d])' % (qb, kb), "")
            elif step <= 6:
                i = 3 + 2 * (step - 1)
                st.asm("v_max3_f32 %%0, %%0, v%d, v%d" % (s + i, s + i + 1), '"+v"(MX[%d][%d])' % (qb, kb), "")
            else:
                st.asm("v_max_f32 %%0, %%0, v%d" % (s + 15), '"+v"(MX[%d][%d])' % (qb, kb), "")
        elif kind == "kpre":
            _, ks, kb = f
            kread(ks, kb, par * TILE)                        # K(t + 2) lives in ring slot t & 1
        elif kind == "dma":
            st.raw("ATTN4_DMA_%s(%d);" % (f[1], f[2]))
        elif kind == "pre":
            emit_prescale(st, f[1], f[2], f[3])
        elif kind == "dec":
            emit_dec_step(st, f[1])
        else:
            raise ValueError(kind)

    # ------------------------------------------------------------------ phase A: 32 gaps
    a_fill = [[] for _ in range(32)]
    a_pre = []                                               # fillers ahead of phase A's first MFMA
    if sm:
        # pair a's scaling sits ONE GAP AHEAD of its two exponentials (one wave per SIMD: nobody covers the latency of a dependent
        # VALU -> transcendental pair issued back to back; measured: +23 % on the whole kernel with the scaling right in front of them)
        kstep, qb, kb, r = elem(0)
        a_pre.append(("scl", kb, qb, r))
        for a in range(32):
            for e in (2 * a, 2 * a + 1):
                kstep, qb, kb, r = elem(e)
                a_fill[a].append(("exp", kb, qb, r))
            if a + 1 < 32:
                kstep, qb, kb, r = elem(2 * a + 2)
                a_fill[a].append(("scl", kb, qb, r))
            if a >= 1:
                a_fill[a].append(("cvt", a - 1))
    if qk:
        for j in range(4):
            for kb in range(2):
                a_fill[4 * j + 5 + kb].append(("kread", 4 + j, kb))
    if head:                                                 # Q~ fragments of d-steps 2..7 are prepared under the first score MFMAs
        for f in range(12):
            for i in range(4):
                a_fill[2 * f + (i >> 1)].append(("pre", f % 2, 2 + f // 2, i))
    if pv:
        for db, g in enumerate((20, 23, 26, 29)):
            a_fill[g].append(("vread", 0, db))
    if not head:                                             # K(t + 2) -> the slot K(t) has left (its last reader passed barrier t - 1)
        for i in range(4):
            a_fill[1 + i].append(("dma", "K", i))

    for f in a_pre:
        do_filler(f)
    counts = []
    for a in range(32):
        n0 = len(st.lines)
        if qk:
            ks, kb, qb = a >> 2, (a >> 1) & 1, a & 1
            if qb == 0 and kb == 0:                          # one wait per d-step: both key blocks' fragments
                st.wait(kread_id.get((ks, 1), 0))
            d = vreg(s_blk(nxt, kb, qb), 16)
            src_c = d if ks else ("0" if head else vreg(nm_blk(qb), 16))
            if "nomfma" not in ABL:
                st.asm("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (d, areg(k_frag(ks, kb), 4), areg(q_frag(qb, ks), 4), src_c))
            st.mfma_pad()
        for f in a_fill[a]:
            do_filler(f)
        counts.append(len(st.lines) - n0)
    st.stats.append(("A", counts))

    # ------------------------------------------------------------------ phase B: 32 gaps
    b_fill = [[] for _ in range(32)]
    if sm:                                                   # row sums: all 64 elements here, two or three per gap (phase A carries the scalings instead)
        e = 0
        for b in range(24):
            for _ in range(2 if (pv and b % 3 == 0) else 3):
                if e < 64:
                    kstep, qb, kb, r = elem(e)
                    b_fill[b].append(("add", kb, qb, r))
                    e += 1
        assert e == 64
    if pv:
        for k in range(1, 4):
            for db in range(4):
                b_fill[8 * (k - 1) + 2 * db].append(("vread", k, db))
    if mx:
        ops = [("max", qb, kb, step) for step in range(8) for qb in range(2) for kb in range(2)]
        if head:
            gaps = sorted(list(range(2, 32)) + [3, 5])       # 32 operations over gaps 2..31
        else:                                                # ... over gaps 2..26; the cross-lane step and the decision mask under gaps 27..31
            gaps = sorted(list(range(2, 27)) + [3, 7, 11, 15, 19, 23, 25])
            for i in range(5):
                b_fill[27 + i].append(("dec", i))
        for op, g in zip(ops, gaps):
            b_fill[g].append(op)
        for ks in range(4):
            for kb in range(2):
                b_fill[24 + ks * 2 + kb].append(("kpre", ks, kb))
    for i in range(4):                                       # V(t + 2) behind the barrier; K(t + 2) went out in phase A (gaps 1..4)
        b_fill[24 + i].append(("dma", "V", i))

    if sm:                                                   # the last pair's packing
        do_filler(("cvt", 31))
    if qk and not pv:                                        # nothing separates the last S MFMA from its first reader: pad the 8-pass latency
        st.asm("s_nop 15\\n\\ts_nop 15")
    counts = []
    for b in range(32):
        n0 = len(st.lines)
        if b == 24 and "nobar" not in ABL:
            st.raw("ATTN4_BARRIER();")
        if pv:
            kstep, db, qb = b >> 3, (b >> 1) & 3, b & 1
            if qb == 0 and db % 2 == 0:                      # one wait per pair of d-blocks
                st.wait(vread_id.get((kstep, db + 1), 0))
            d = areg(o_blk(db, qb), 16)
            if "nomfma" not in ABL:
                st.asm("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (d, areg(v_frag(kstep, db), 4), vreg(p_frag(kstep, qb), 4), d))
            st.mfma_pad()
        if b == 1 and mx:
            emit_mask(st, nxt)
        for f in b_fill[b]:
            do_filler(f)
        counts.append(len(st.lines) - n0)
    st.stats.append(("B", counts))
    if mx and not ("nodecide" in ABL and not head):
        emit_decide(st, nxt, head)
    return st


def gen_zero_o():
    st = Stream(False)
    for r in range(128):
        st.asm("v_accvgpr_write_b32 a%d, 0" % (O_A + r))
    return st


def gen_epi_read(qb, db):
    """x_[r] = O[db][qb][r] into compiler-allocated floats (16 at a time: hipcc owns 64 registers only)"""
    st = Stream(False)
    for r in range(16):
        st.asm("v_accvgpr_read_b32 %%0, a%d" % (o_blk(db, qb) + r), '"=v"(x_[%d])' % r, "")
    return st


QRAW_V = S_V + 64                                            # raw Q rows land in the registers of score tile 1 (first written in step 0)


def q_raw(qb, ks):
    return QRAW_V + 4 * (qb * 8 + ks)


def emit_prescale(st, qb, ks, i):
    """word i of fragment (qb, ks): the two bf16 as stored -> accumulator register (no arithmetic: the softmax scale is applied in fp32, see header)"""
    v, a = q_raw(qb, ks) + i, q_frag(qb, ks) + i
    st.asm("v_accvgpr_write_b32 a%d, v%d" % (a, v))


def gen_q_load():
    """the wave's 64 query rows, 16 B per lane and fragment: two base pointers (qp0_, qp1_), immediates over d"""
    st = Stream(False)
    for qb in range(2):
        text = "\\n\\t".join("global_load_dwordx4 %s, %%0, off offset:%d" % (vreg(q_raw(qb, ks), 4), ks * 32) for ks in range(8))
        st.asm(text, "", '"v"(qp%d_)' % qb, '"memory"')
    return st


def gen_q_pre01():
    st = Stream(False)
    for ks in range(2):
        for qb in range(2):
            for i in range(4):
                emit_prescale(st, qb, ks, i)
    return st


def main():
    global OUT
    if "--abl" in sys.argv:
        names = sys.argv[sys.argv.index("--abl") + 1]
        ABL.update(names.split(","))
        # ablation streams are scratch: build/<dir>/ (git-ignored), never next to the product streams
        OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "attn4_abl", "attn4_gen_" + names.replace(",", "_"))
    os.makedirs(OUT, exist_ok=True)
    report = []
    for safe in (False, True):
        for mode, pars in (("head", (1,)), ("loop", (0, 1)), ("tail", (0, 1))):
            for par in pars:
                st = gen(mode, par, safe)
                name = "%s%s%s.inc" % ("safe_" if safe else "", mode, "" if mode == "head" else str(par))
                with open(os.path.join(OUT, name), "w") as f:
                    f.write("\n".join(st.lines) + "\n")
                if not safe:
                    for ph, c in st.stats:
                        report.append("%-5s p%d phase %s: statements per gap %s  total %d" % (mode, par, ph, " ".join(map(str, c)), sum(c)))
                    report.append("%-5s p%d: most LDS reads outstanding (if none completed unwaited) %d" % (mode, par, st.max_out))
    with open(os.path.join(OUT, "zero_o.inc"), "w") as f:
        f.write("\n".join(gen_zero_o().lines) + "\n")
    with open(os.path.join(OUT, "epi_read.inc"), "w") as f:
        for qb in range(2):
            for db in range(4):
                f.write("if (qb_ == %d && db_ == %d) {\n%s\n}\n" % (qb, db, "\n".join(gen_epi_read(qb, db).lines)))
    with open(os.path.join(OUT, "q_load.inc"), "w") as f:
        f.write("\n".join(gen_q_load().lines) + "\n")
    with open(os.path.join(OUT, "q_pre01.inc"), "w") as f:
        f.write("\n".join(gen_q_pre01().lines) + "\n")
    print("\n".join(report))


if __name__ == "__main__":
    main()

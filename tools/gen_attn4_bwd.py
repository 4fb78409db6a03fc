#!/usr/bin/env python3
"""Generator of the hand-placed instruction streams of the d == 128 attention BACKWARD kernels (metamorph_amd/csrc/attn4_bwd.hip).

Same construction as tools/gen_attn4.py (one wave per SIMD, one `asm volatile` per instruction on literal registers, hipcc confined to
v[0:63]); read that header first.  Both backward kernels are the same machine:

    a wave keeps ONE side of the products in the accumulator file for its whole life (the "persistent" B operands) and streams tiles of
    the other side through LDS (a four-slot ring filled by LDS-DMA).  Per tile:
      phase A   32 MFMAs:  X(t+1) = Xsrc . P0^T  and  Y(t+1) = Ysrc . P1^T   (A operands: ds_read_b128 fragments of the tile)
                ||  P = exp2(c X(t)),  dS = P * Y(t),  packing to bf16        (X / Y arrive already shifted: C = -lse / scale, -delta;
                                                                                c = scale * log2 e enters in fp32: one v_mul_f32 per score)
      phase B   the gradient products of tile t, A operands gathered TRANSPOSED from the same tile (ds_read_b64_tr_b16 pairs)

    kind 'kv' (dK / dV):  persistent K (as stored) and V of the wave's 32 keys; tiles = 64 query rows of (Q, dO, -lse / scale, -delta);
                          X = S [q][key], Y = dP;  phase B: dV^T += dO^T P (16 MFMAs), dK^T += Q^T dS (16 MFMAs); the C operands of the
                          score chains are pre-loaded from the tile's statistics rows straight into the chain's registers (ds_read_b128)
    kind 'q'  (dQ):       persistent Q (as stored) and dO of the wave's 32 query rows (+ two 16-register tuples -lse / scale, -delta as first C operands);
                          tiles = 64 keys of (K, V);  X = S^T [key][q], Y = dP^T;  phase B: dQ^T += K^T dS^T (16 MFMAs)

Register map (a = accumulator file):
    kv:  a[0:63] dK^T | a[64:127] dV^T | a[128:159] K~ | a[160:191] V | a[192:223] b128 ring | a[224:255] transposed ring
         v[96:223] X / Y of two tiles | v[224:239] P fragments | v[240:255] dS fragments          (hipcc: v[0:95])
    q:   a[0:63] dQ^T | a[64:95] Q~ | a[96:127] dO | a[128:159] b128 ring | a[160:191] transposed ring
         v[80:207] X / Y of two tiles | v[208:223] dS fragments | v[224:239] -lse2 tuple | v[240:255] -delta tuple   (hipcc: v[0:79])

    python tools/gen_attn4_bwd.py          # rewrites metamorph_amd/csrc/attn4_bwd_gen/{kv,q}_*.inc

Arithmetic (round 5, as in the forward stream): the score chains run on q and k AS STORED, X = q . k - lse / scale in fp32, and the softmax
scale enters on the fp32 side, P = exp2((scale * log2 e) * X).  Rounds 1-4 kept a re-rounded bf16 copy of the persistent operand multiplied
by scale * log2 e (K~ in the dK / dV kernel, Q~ in the dQ kernel -- so the two kernels and the forward recomputed three slightly different
P from the same q, k): one rounding more than the reference has, with a score error growing with |s|.
"""
import os
import sys

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "metamorph_amd", "csrc", "attn4_bwd_gen")
ABL = set()


def areg(lo, n):
    return "a[%d:%d]" % (lo, lo + n - 1)


def vreg(lo, n):
    return "v[%d:%d]" % (lo, lo + n - 1)


class Kind:
    def __init__(self, name):
        self.name = name
        kv = name == "kv"
        self.kv = kv
        self.NB = 32 if kv else 16                           # MFMAs of phase B
        self.ACC0 = 0                                        # dK^T / dQ^T
        self.ACC1 = 64                                       # dV^T (kv)
        self.PERS0 = 128 if kv else 64
        self.PERS1 = 160 if kv else 96
        self.RINGA = 192 if kv else 128
        self.RINGT = 224 if kv else 160
        self.XY = 96 if kv else 80                           # hipcc owns v[0:XY-1] (amdgpu_num_vgpr)
        self.PK0 = 224                                       # P fragments (kv)
        self.PK1 = 240 if kv else 208                        # dS fragments
        self.NL = 224                                        # q: -lse2 tuple
        self.ND = 240                                        # q: -delta tuple
        self.SLOT = 32768                                    # kv: Q tile | dO tile;  q: K tile | V tile  (kv: the statistics rows have their own 4 x 512 B ring)
        self.NDMA = 9 if kv else 8                           # LDS-DMA pieces per wave and tile
        self.BAR = 24 if kv else 8                           # phase-B gap of the step's barrier

    def x(self, par, i):
        return self.XY + 16 * (par * 4 + i)

    def y(self, par, i):
        return self.XY + 16 * (par * 4 + 2 + i)


class Stream:
    def __init__(self, safe):
        self.safe = safe
        self.lines = []
        self.issued = 0
        self.done = 0
        self.stats = []
        self.max_out = 0

    def raw(self, text):
        self.lines.append(text)

    def asm(self, text, outs="", ins="", clobbers=""):
        # gfx950: a VALU instruction that reads the result of the transcendental instruction issued right before it needs one wait
        # state (hipcc pads nothing inside asm); in the pipelined steps an MFMA always sits in between, in the tail nothing does
        import re
        if self.lines:
            m = re.match(r'asm volatile\("v_exp_f32 v(\d+),', self.lines[-1])
            if m and re.search(r"\bv%s\b" % m.group(1), text.split(",", 1)[-1]):
                self.lines.append('asm volatile("s_nop 0");')
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
    """VALU order = consumption order of the packed fragments: (kstep, block i, register r)"""
    kstep, m = e >> 3, e & 7
    return kstep, kstep >> 1, 8 * (kstep & 1) + m


def gen(K, mode, u4, safe):
    """mode: 'head' (u = -1) | 'loop' | 'tail';  u4 = u & 3 (tile u sits in ring slot u4, its X / Y in buffer u4 & 1)"""
    head, tail = mode == "head", mode == "tail"
    if head:
        u4 = 3                                               # u = -1
    par = u4 & 1
    nxt = par ^ 1
    s_cur, s_n1, s_n2 = u4, (u4 + 1) & 3, (u4 + 2) & 3        # ring slots of tiles u, u + 1, u + 2
    qk = not tail                                            # phase-A products of tile u + 1
    va = not head                                            # VALU work + phase-B products of tile u
    st = Stream(safe)
    st.raw("// ---- generated by tools/gen_attn4_bwd.py: kind %s, %s, u & 3 = %d%s ----" % (K.name, mode, u4, ", SAFE (serialised)" if safe else ""))

    # a DS immediate spans 64 KiB: ring slots 0, 1 are addressed from the lane constants RA / TA / TB, slots 2, 3 from their copies + 64 KiB
    def a_read(a, slot):
        """b128 fragment of phase-A MFMA a (of the tile in ring slot `slot`) -> ring A"""
        ks, j = a >> 2, a & 3
        off = (slot & 1) * K.SLOT + (16384 if j >= 2 else 0) + (j & 1) * 8192
        r = K.RINGA + 4 * (a % 8)
        st.asm("ds_read_b128 %s, %%0 offset:%d" % (areg(r, 4), off), "", '"v"(%s[%d])' % ("RAH" if slot >= 2 else "RA", ks))
        return st.lds()

    def t_read(b, slot):
        """transposed fragment of phase-B MFMA b -> ring T (two b64 gathers: rows +0 and +8 of the 16-row step)"""
        if K.kv:
            kstep, db, j = b >> 3, (b >> 1) & 3, b & 1
            part = 16384 if j == 0 else 0                    # dV^T takes dO^T, dK^T takes Q^T
        else:
            kstep, db = b >> 2, b & 3
            part = 0                                         # K^T
        off = (slot & 1) * K.SLOT + part + kstep * 4096
        r = K.RINGT + 4 * (b % 8)
        h = "H" if slot >= 2 else ""
        st.asm("ds_read_b64_tr_b16 %s, %%0 offset:%d\\n\\tds_read_b64_tr_b16 %s, %%1 offset:%d" % (areg(r, 2), off, areg(r + 2, 2), off),
               "", '"v"(TA%s[%d]), "v"(TB%s[%d])' % (h, db, h, db))
        return st.lds(2)

    def preload(idx, slot, par_dst):
        """kv: 4 statistics values (-lse2 or -delta of 4 consecutive query rows) -> 4 registers of a score chain's accumulator"""
        which, i, q4 = idx >> 3, (idx >> 2) & 1, idx & 3      # which: 0 = X (-lse2), 1 = Y (-delta); block i; rows 8*q4 + 4*hi ..
        base = (K.x if which == 0 else K.y)(par_dst, i) + 4 * q4
        off = slot * 512 + which * 256 + (i * 32 + 8 * q4) * 4
        st.asm("ds_read_b128 %s, %%0 offset:%d" % (vreg(base, 4), off), "", '"v"(SA)')
        return st.lds()

    aread_id, tread_id, pre_id = {}, {}, []

    # ---- entry state: what the previous step left in flight
    if head:
        if K.kv:
            for idx in range(16):
                pre_id.append(preload(idx, s_n1, nxt))
        for a in range(8):
            aread_id[a] = a_read(a, s_n1)
    else:
        if K.kv and qk:
            for idx in range(16):
                st.issued += 1
                pre_id.append(st.issued)
        if qk:
            for a in range(8):
                st.issued += 1
                aread_id[a] = st.issued
        else:                                                # tail: the previous step still issued them (for a tile nobody computes)
            st.issued += 8 + (16 if K.kv else 0)
        if safe:
            st.done = st.issued

    # ------------------------------------------------------------------ phase A
    if va:
        st.raw("if (mask_cur) {")
        for i in range(2):
            for r in range(16):
                c = i * 32 + 8 * (r >> 2) + (r & 3)
                v = K.x(par, i) + r
                if K.kv:                                     # visible: (row - lo) <u span
                    st.asm("v_sub_u32 %%0, %d, %%1\\n\\tv_cmp_gt_u32 vcc, %%2, %%0\\n\\ts_nop 1\\n\\tv_cndmask_b32 v%d, %%3, v%d, vcc" % (c, v, v),
                           '"=&v"(t0_)', '"v"(mlo_), "v"(mspan_), "v"(ninf)', '"vcc"')
                else:                                        # masked: key > lim
                    st.asm("v_cmp_gt_i32 vcc, %d, %%0\\n\\ts_nop 1\\n\\tv_cndmask_b32 v%d, v%d, %%1, vcc" % (c, v, v), "", '"v"(mlim_), "v"(ninf)', '"vcc"')
        st.raw("}")
    a_fill = [[] for _ in range(32)]
    NB = K.NB
    b_fill = [[] for _ in range(NB)]
    b_pre = []                                               # phase-B fillers ahead of its first MFMA
    a_pre = []                                               # phase-A fillers ahead of its first MFMA
    if va:
        # element e (consumption order of the packed fragments): exp, one gap later mul, a pair's packing behind its second element.
        # The first SPLIT elements sit in phase A, the rest under the first MFMAs of phase B (their fragments are consumed last).
        SPLIT = 16 if K.kv else 24
        slot_of = {}
        for e in range(SPLIT):
            slot_of[e] = ("A", (e * 32) // SPLIT)
        for e in range(SPLIT, 32):
            slot_of[e] = ("B", e - SPLIT)

        def put(ph, g, f):
            if ph == "A" and g >= 32:
                ph, g = "B", g - 32
            if ph == "B" and g < 0:
                b_pre.append(f)
            elif ph == "A":
                a_fill[g].append(f)
            else:
                b_fill[min(g, NB - 1)].append(f)

        for e in range(32):
            ph, g = slot_of[e]
            if e % 2 == 0:
                # the pair's scaling (raw q . k units -> log2 domain) ONE SLOT AHEAD of its exponentials: one wave per SIMD, nobody covers the
                # latency of a dependent VALU -> transcendental pair issued back to back (measured: +12 % on the kernels)
                if ph == "A" and g == 0:
                    a_pre.append(("scl", e >> 1))
                elif ph == "B" and g == 0:
                    a_fill[31].append(("scl", e >> 1))
                else:
                    put(ph, g - 1, ("scl", e >> 1))
            put(ph, g, ("exp", e))
            put(ph, g + 1, ("mul", e))
        for p in range(16):
            ph, g = slot_of[2 * p + 1]
            if K.kv:
                put(ph, g + 1, ("cvt0", p))
            put(ph, g + 2, ("cvt1", p))
        for b in range(8):
            a_fill[24 + b].append(("tread", b))
    if qk:
        for a in range(8, 32):
            a_fill[a - 8].append(("aread", a))
        for i in range(K.NDMA):
            a_fill[1 + i].append(("dma", i))

    def do_filler(f):
        kind = f[0]
        if ("no" + kind.rstrip("01")) in ABL or (kind in ("aread", "tread", "apre", "preload") and "nolds" in ABL):
            return
        if kind == "exp":
            kstep, i, r = elem(f[1])
            v = K.x(par, i) + r
            st.asm("v_exp_f32 v%d, v%d" % (v, v))
        elif kind == "scl":                                  # two plain multiplies (a v_pk_mul_f32 costs ~22 cycles between MFMAs: tools/gen_attn4.py header)
            kstep, i, r = elem(2 * f[1])
            v = K.x(par, i) + r
            st.asm("v_mul_f32 v%d, %%0, v%d\\n\\tv_mul_f32 v%d, %%0, v%d" % (v, v, v + 1, v + 1), "", '"s"(sl2b_)')
        elif kind == "mul":
            kstep, i, r = elem(f[1])
            st.asm("v_mul_f32 v%d, v%d, v%d" % (K.y(par, i) + r, K.y(par, i) + r, K.x(par, i) + r))
        elif kind in ("cvt0", "cvt1"):
            p = f[1]
            kstep, i, r = elem(2 * p)
            src = (K.x if kind == "cvt0" else K.y)(par, i) + r
            dst = (K.PK0 if kind == "cvt0" else K.PK1) + 4 * kstep + (p & 3)
            st.asm("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (dst, src, src + 1))
        elif kind == "aread":
            aread_id[f[1]] = a_read(f[1], s_n1)
        elif kind == "tread":
            tread_id[f[1]] = t_read(f[1], s_cur)
        elif kind == "apre":
            a_read(f[1], s_n2)
        elif kind == "preload":
            preload(f[1], s_n2, par)
        elif kind == "dma":
            st.raw("BWD_DMA(%d);" % f[1])
        else:
            raise ValueError(kind)

    for f in a_pre:
        do_filler(f)
    counts = []
    for a in range(32):
        n0 = len(st.lines)
        if qk:
            ks, j = a >> 2, a & 3
            if a == 0 and pre_id:
                st.wait(pre_id[-1])
            st.wait(aread_id.get(a, 0))
            d = vreg((K.x if j < 2 else K.y)(nxt, j & 1), 16)
            if ks == 0 and not K.kv:
                c = vreg(K.NL if j < 2 else K.ND, 16)
            else:
                c = d
            pers = (K.PERS0 if j < 2 else K.PERS1) + 4 * ks
            if "nomfma" not in ABL:
                st.asm("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (d, areg(K.RINGA + 4 * (a % 8), 4), areg(pers, 4), c))
            st.mfma_pad()
        for f in a_fill[a]:
            do_filler(f)
        counts.append(len(st.lines) - n0)
    st.stats.append(("A", counts))

    # ------------------------------------------------------------------ phase B
    if va:
        for b in range(8, NB):
            b_fill[b - 8].append(("tread", b))
    if not tail:                                             # behind the barrier: tile u + 2 is on chip -> C operands and first fragments of the next step
        if K.kv:
            for idx in range(16):
                b_fill[K.BAR + idx // 2].append(("preload", idx))
        for a in range(8):
            b_fill[K.BAR + a].append(("apre", a))
    for f in b_pre:
        do_filler(f)
    if qk and not va:                                        # head: nothing separates the last MFMA of phase A from what follows; pad its latency
        st.asm("s_nop 15\\n\\ts_nop 15")
    counts = []
    for b in range(NB):
        n0 = len(st.lines)
        if b == K.BAR and not tail and "nobar" not in ABL:
            st.raw("BWD_BARRIER();")
        if va:
            st.wait(tread_id.get(b, 0))
            if K.kv:
                kstep, db, j = b >> 3, (b >> 1) & 3, b & 1
                acc = (K.ACC1 if j == 0 else K.ACC0) + 16 * db
                pk = (K.PK0 if j == 0 else K.PK1) + 4 * kstep
            else:
                kstep, db = b >> 2, b & 3
                acc = K.ACC0 + 16 * db
                pk = K.PK1 + 4 * kstep
            if "nomfma" not in ABL:
                st.asm("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (areg(acc, 16), areg(K.RINGT + 4 * (b % 8), 4), vreg(pk, 4), areg(acc, 16)))
            st.mfma_pad()
        for f in b_fill[b]:
            do_filler(f)
        counts.append(len(st.lines) - n0)
    st.stats.append(("B", counts))
    return st


def gen_zero(K):
    st = Stream(False)
    for r in range(128 if K.kv else 64):
        st.asm("v_accvgpr_write_b32 a%d, 0" % r)
    return st


def gen_acc_read(K):
    """x_[r] = accumulator block (which_, db_)[4 * i4_ + r], four at a time (hipcc owns few registers: keep its epilogue small)"""
    out = []
    for which in range(2 if K.kv else 1):
        for db in range(4):
            for i4 in range(4):
                st = Stream(False)
                base = (K.ACC0 if which == 0 else K.ACC1) + 16 * db + 4 * i4
                for r in range(4):
                    st.asm("v_accvgpr_read_b32 %%0, a%d" % (base + r), '"=v"(x_[%d])' % r, "")
                out.append("if (which_ == %d && db_ == %d && i4_ == %d) {\n%s\n}" % (which, db, i4, "\n".join(st.lines)))
    return "\n".join(out)


def gen_pers_load(K):
    """the 16 row fragments a lane keeps (operand 0 from pointer p0_, operand 1 from p1_; 16 B each, immediates over d) -> the registers
    of the first X / Y buffers (free until the head): hipcc never holds them (it owns few registers and would park them in the
    accumulator file, which is the streams')"""
    st = Stream(False)
    for p in range(2):
        text = "\\n\\t".join("global_load_dwordx4 %s, %%0, off offset:%d" % (vreg(K.XY + 32 * p + 4 * ks, 4), ks * 32) for ks in range(8))
        st.asm(text, "", '"v"(p%d_)' % p, '"memory"')
    return st


def gen_pers_place(K):
    """both persistent operands as stored -> their accumulator registers (no arithmetic: the softmax scale is applied in fp32, see header)"""
    st = Stream(False)
    for p in range(2):
        for ks in range(8):
            for i in range(4):
                v = K.XY + 32 * p + 4 * ks + i
                a = (K.PERS0 if p == 0 else K.PERS1) + 4 * ks + i
                st.asm("v_accvgpr_write_b32 a%d, v%d" % (a, v))
    return st


def gen_tuple_write(K):
    """q: -lse / scale and -delta of the lane's query row -> all 16 registers of the two C tuples"""
    st = Stream(False)
    for r in range(16):
        st.asm("v_mov_b32 v%d, %%0" % (K.NL + r), "", '"v"(nl_)')
        st.asm("v_mov_b32 v%d, %%0" % (K.ND + r), "", '"v"(nd_)')
    return st


def main():
    global OUT
    if "--abl" in sys.argv:
        names = sys.argv[sys.argv.index("--abl") + 1]
        ABL.update(names.split(","))
        # ablation streams are scratch: build/<dir>/ (git-ignored), never next to the product streams
        OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "attn4_abl", "attn4_bwd_gen_" + names.replace(",", "_"))
    os.makedirs(OUT, exist_ok=True)
    report = []
    for kname in ("kv", "q"):
        K = Kind(kname)
        for safe in (False, True):
            for mode, us in (("head", (3,)), ("loop", (0, 1, 2, 3)), ("tail", (0, 1, 2, 3))):
                for u4 in us:
                    st = gen(K, mode, u4, safe)
                    name = "%s_%s%s%s.inc" % (kname, "safe_" if safe else "", mode, "" if mode == "head" else str(u4))
                    with open(os.path.join(OUT, name), "w") as f:
                        f.write("\n".join(st.lines) + "\n")
                    if not safe and u4 in (0, 3):
                        for ph, c in st.stats:
                            report.append("%-2s %-5s u&3=%d phase %s: statements per gap %s  total %d" % (kname, mode, u4, ph, " ".join(map(str, c)), sum(c)))
                        report.append("%-2s %-5s u&3=%d: most LDS reads outstanding (if none completed unwaited) %d" % (kname, mode, u4, st.max_out))
        with open(os.path.join(OUT, "%s_zero.inc" % kname), "w") as f:
            f.write("\n".join(gen_zero(K).lines) + "\n")
        with open(os.path.join(OUT, "%s_acc_read.inc" % kname), "w") as f:
            f.write(gen_acc_read(K) + "\n")
        with open(os.path.join(OUT, "%s_pers_load.inc" % kname), "w") as f:
            f.write("\n".join(gen_pers_load(K).lines) + "\n")
        with open(os.path.join(OUT, "%s_pers_place.inc" % kname), "w") as f:
            f.write("\n".join(gen_pers_place(K).lines) + "\n")
        if not K.kv:
            with open(os.path.join(OUT, "q_tuple_write.inc"), "w") as f:
                f.write("\n".join(gen_tuple_write(K).lines) + "\n")
    print("\n".join(report))


if __name__ == "__main__":
    main()

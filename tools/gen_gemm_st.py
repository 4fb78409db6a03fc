#!/usr/bin/env python3
"""Generator of the hand-placed instruction streams of csrc/gemm_st.hip: the one-wave-per-SIMD bf16 GEMM (256 x 256 x 64 block tile, four
waves, each a 128 x 128 wave tile = 8 x 8 blocks of v_mfma_f32_16x16x32_bf16 with ALL 256 accumulator registers of the wave in a[0:255]).

One K stage (64 k = two sub-steps u = 0, 1 of 32 k) is 128 MFMAs = 2048 matrix-pipe cycles per SIMD.  Around them, per stage t (ring slot
p = t & 1 of two 64-KiB stage images [X 256 rows | W 256 rows] x 128 B, 16-B chunk ^ (row & 7)):

    sub-step 0 (fragments of (t, 0) in buffer 0)      sub-step 1 (fragments of (t, 1) in buffer 1)
    gaps RD1 ..      16 ds_read_b128: (t, 1) -> buffer 1   gaps .. : the rest of the LDS-DMA pieces of stage t + 2
    gap B1           lgkmcnt(0), s_barrier  [slot p free]  gap B2    vmcnt(pieces of t + 2 issued so far), s_barrier  [stage t + 1 landed]
    gaps B1+1 ..     LDS-DMA pieces of stage t + 2 -> slot p  gaps B2+1 .. 16 ds_read_b128: (t + 1, 0) -> buffer 0 from slot p ^ 1

The accumulator block (jn, im) holds the TRANSPOSED product (rows = 16 output columns n, lane & 15 = output row m), i.e. the W fragment
is the MFMA's A operand and the X fragment its B operand: a lane then owns four ADJACENT output columns of one row and the epilogue
drains a block with one ds_write_b128 straight from the accumulator registers.  Summation order over k is that of gemm_pp_tile (one
16x16x32 MFMA per 32 k, ascending): results are bit-identical to variant 11.

Registers: hipcc v[0:HV-1] (amdgpu_num_vgpr(HV)); fragments v[HV + 64 b + 4 jn] (W, b = buffer) and v[HV + 64 b + 32 + 4 im] (X).

    python tools/gen_gemm_st.py            -> metamorph_amd/csrc/gemm_st_gen/*.inc
    python tools/gen_gemm_st.py --abl nodma,nolds,nobar   -> build/gemm_st_abl_<name>/ (timing-only streams, wrong results)
"""
import argparse
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HV = 96                       # first stream VGPR
NP = 16                       # LDS-DMA pieces per wave and stage (8 X + 8 W, 1 KiB each)


def acc(jn, im):
    b = 4 * (jn * 8 + im)
    return f"a[{b}:{b + 3}]"


def wfrag(buf, jn):
    b = HV + 64 * buf + 4 * jn
    return f"v[{b}:{b + 3}]"


def xfrag(buf, im):
    b = HV + 64 * buf + 32 + 4 * im
    return f"v[{b}:{b + 3}]"


# order in which the 16 fragments of a sub-step are read = order in which the MFMA walk (jn outer, im inner) first needs them
READ_ORDER = [("W", 0)] + [("X", i) for i in range(8)] + [("W", j) for j in range(1, 8)]


def read_line(kind, idx, buf, slot, u):
    dst = wfrag(buf, idx) if kind == "W" else xfrag(buf, idx)
    arr = "WA" if kind == "W" else "XA"
    return f'asm volatile("ds_read_b128 {dst}, %0 offset:{idx * 2048}" :: "v"({arr}[{slot}][{u}]));'


class Opts:
    def __init__(self, **kw):
        self.rd1 = 8          # first gap of the buffer-1 reads (sub-step 0)
        self.b1 = 32          # gap of barrier 1 (sub-step 0)
        self.b2 = 40          # gap of barrier 2 (sub-step 1)
        self.dma_every = 5    # one LDS-DMA piece every this many gaps after barrier 1
        self.abl = set()
        self.safe = False
        self.stagger = False  # wave k issues its piece k gaps later
        self.__dict__.update(kw)


def stage(o, p, first):
    """one K stage on ring slot p; first: the tile's first sub-step (C = 0)"""
    L = []
    say = L.append
    full = 'asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");'
    # DMA schedule: piece i after gap g of the stage-long gap index (0..127)
    dma_at = {}
    g = o.b1 + 1
    for i in range(NP):
        dma_at[g] = i
        g += o.dma_every
    assert g - o.dma_every + (3 if o.stagger else 0) < 128, "DMA pieces run past the stage"
    issued_before_b2 = sum(1 for gg in dma_at if gg + (3 if o.stagger else 0) <= 64 + o.b2 and not ("dmahalf" in o.abl and dma_at[gg] % 2))
    for u in range(2):
        buf = u
        for g in range(64):
            jn, im = divmod(g, 8)
            if u == 0 and "nolds" not in o.abl:
                # buffer 0 was read in gaps b2+1 .. b2+16 of the previous sub-step 1, in READ_ORDER: progressive waits, then all
                if g < 8:
                    say(f'asm volatile("s_waitcnt lgkmcnt({14 - g})");')
                elif g == 8:
                    say('asm volatile("s_waitcnt lgkmcnt(0)");')
            c = "0" if (first and u == 0) else acc(jn, im)
            if "nomfma" not in o.abl:
                say(f'asm volatile("v_mfma_f32_16x16x32_bf16 {acc(jn, im)}, {wfrag(buf, jn)}, {xfrag(buf, im)}, {c}");')
            gg = u * 64 + g
            # ---- fillers behind MFMA g
            if u == 0 and o.rd1 <= g < o.rd1 + 16 and "nolds" not in o.abl:
                kind, idx = READ_ORDER[g - o.rd1]
                say(read_line(kind, idx, 1, p, 1))
                if o.safe:
                    say(full)
            if u == 0 and g == o.b1:
                say('asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");')
                if "nobar" not in o.abl:
                    say("GST_BARRIER();")
            if o.stagger:
                for k in range(4):                           # wave k issues piece i one gap after wave k - 1 (the four share one TA)
                    if gg - k in dma_at and "nodma" not in o.abl:
                        say(f"GST_DMA_IF({dma_at[gg - k]}, {p}, {k});")
                        if o.safe:
                            say(full)
            elif gg in dma_at and "nodma" not in o.abl and not ("dmahalf" in o.abl and dma_at[gg] % 2):
                say(f"GST_DMA({dma_at[gg]}, {p});")
                if o.safe:
                    say(full)
            if u == 1 and g == o.b2:
                if "nodma" not in o.abl and "nowaitvm" not in o.abl:
                    say(f'asm volatile("s_waitcnt vmcnt({0 if o.safe else issued_before_b2})" ::: "memory");')
                if "nobar" not in o.abl:
                    say("GST_BARRIER();")
            if u == 1 and o.b2 < g <= o.b2 + 16 and "nolds" not in o.abl:
                kind, idx = READ_ORDER[g - o.b2 - 1]
                say(read_line(kind, idx, 0, p ^ 1, 0))
                if o.safe:
                    say(full)
    assert o.rd1 + 16 <= o.b1 and o.b2 + 16 < 64
    return L


def trip(o, first):
    L = [f"// ---- generated by tools/gen_gemm_st.py: one trip = K stages on ring slots 0, 1{' (first of a tile: C = 0)' if first else ''}"
         f"{' [serialised]' if o.safe else ''} ----"]
    L += stage(o, 0, first)
    L.append("GST_NEXT_STAGE();")
    L += stage(o, 1, False)
    return L


def pro_reads(o):
    L = ["// ---- generated by tools/gen_gemm_st.py: fragments of (stage 0, sub-step 0) -> buffer 0 ----"]
    for kind, idx in READ_ORDER:
        L.append(read_line(kind, idx, 0, 0, 0))
    return L


def drain(o):
    """accumulator blocks (jn = 0..7, im) -> the wave's fp32 staging slab [16 m][128 n] (512-B rows, 16-B chunk ^ (m & 7)): one
    ds_write_b128 per block from the accumulator registers; SW[b] = lane address for jn = 2 j + b, + 128 j"""
    L = ["// ---- generated by tools/gen_gemm_st.py: drain of the m block im_ (wave-uniform) ----", "switch (im_) {"]
    for im in range(8):
        L.append(f"case {im}:")
        for jn in range(8):
            L.append(f'    asm volatile("ds_write_b128 %0, {acc(jn, im)} offset:{(jn >> 1) * 128}" :: "v"(SW[{jn & 1}]) : "memory");')
        L.append("    break;")
    L.append("}")
    return L


def write(d, name, lines):
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, name), "w") as f:
        f.write("\n".join(lines) + "\n")


def emit(d, **kw):
    for safe in (False, True):
        o = Opts(safe=safe, **kw)
        pre = "safe_" if safe else ""
        write(d, pre + "trip_first.inc", trip(o, True))
        write(d, pre + "trip.inc", trip(o, False))
    o = Opts(**kw)
    write(d, "pro_reads.inc", pro_reads(o))
    write(d, "drain.inc", drain(o))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--abl", default="")
    ap.add_argument("--rd1", type=int, default=8)
    ap.add_argument("--b1", type=int, default=32)
    ap.add_argument("--b2", type=int, default=40)
    ap.add_argument("--dma-every", type=int, default=5)
    ap.add_argument("--stagger", action="store_true")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    kw = dict(rd1=a.rd1, b1=a.b1, b2=a.b2, dma_every=a.dma_every, stagger=a.stagger)
    if a.abl:
        for name in a.abl.split(","):
            emit(os.path.join(REPO, "build", "gemm_st_abl_" + name.replace("+", "_")), abl=set(name.split("+")), **kw)
    else:
        emit(a.out or os.path.join(REPO, "metamorph_amd", "csrc", "gemm_st_gen"), **kw)

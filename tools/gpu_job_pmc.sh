#!/bin/bash
# usage: gpu_job_pmc.sh <script.py> ; two PMC passes (SQ set, then LDS/TCC set) + kernel trace; prints per-kernel averages
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp PYTHONUNBUFFERED=1
S=${1:-tools/prof_attn.py}
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|TCC_[A-Z_0-9]*\|GRBM_[A-Z_]*" | sort -u | tr '\n' ' ' | head -c 3000 > gpurun_out/pmc/counters.txt
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pmc -o kt -- python $S > gpurun_out/pmc/kt.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES --output-format csv -d gpurun_out/pmc -o p1 -- python $S > gpurun_out/pmc/p1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU --output-format csv -d gpurun_out/pmc -o p2 -- python $S > gpurun_out/pmc/p2.log 2>&1
ls gpurun_out/pmc
python - <<'PY'
import csv, glob, collections
for tag in ("kt","p1","p2"):
    for f in glob.glob(f"gpurun_out/pmc/**/*{tag}*counter_collection.csv", recursive=True)+glob.glob(f"gpurun_out/pmc/**/*{tag}*kernel_trace.csv", recursive=True):
        rows=list(csv.DictReader(open(f)))
        if not rows: continue
        print("==",f,len(rows)); 
        if "Counter_Name" in rows[0]:
            acc=collections.defaultdict(lambda: collections.defaultdict(list))
            for r in rows:
                k=r["Kernel_Name"][:60]
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k,d in acc.items():
                if "attn" in k or "gemm" in k:
                    print(k, {c: round(sum(v)/len(v)) for c,v in d.items()})
        else:
            acc=collections.defaultdict(list)
            for r in rows:
                acc[r["Kernel_Name"][:60]].append(float(r["End_Timestamp"])-float(r["Start_Timestamp"]))
            for k,v in acc.items():
                if "attn" in k or "gemm" in k: print(k, len(v), "avg_us", round(sum(v)/len(v)/1e3,1))
PY

# SQ counter passes over the trace kernels of the cfg3 step (bench_paths.py, a 2e7-rank window of the
# order-2 candidate space x 16 TX x 64 RX).  Counters only (no other trace domain) -- one pass per set.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_trace && mkdir -p gpurun_out/pmc_trace
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_trace -o $tag -- python bench_paths.py --ranks 20000000 --steps 1 --no-cpu > gpurun_out/pmc_trace/$tag.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc_trace -o ktrace -- python bench_paths.py --ranks 20000000 --steps 1 --no-cpu > gpurun_out/pmc_trace/ktrace.log 2>&1
python - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob('gpurun_out/pmc_trace/**/*counter_collection.csv', recursive=True)):
    vals=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'trace_' in k:
            vals[(k[:60], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in sorted(vals.items()): print(k, sum(v)/len(v), len(v))
PY

# Round-2 profile set (run on the MI355X box from the repo root): kernel trace + stats of the default
# bench, FETCH_SIZE / WRITE_SIZE passes and SQ counter passes of the dense kernel (counters only, no
# other trace domain).  Outputs under gpurun_out/prof_r02; copy the CSVs into profiles/r02/ and run
# `python profiles/summarize.py profiles/r02`.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_r02 && mkdir -p gpurun_out/prof_r02
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02 -o r02 -- python bench.py --no-cpu-baseline > gpurun_out/prof_r02/bench_traced.json 2> gpurun_out/prof_r02/bench_traced.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/prof_r02 -o r02_pmc_$(echo $c | tr A-Z a-z) -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-paths --no-scaling > /dev/null 2>&1
done
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_WR GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/prof_r02 -o sq_$tag -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-paths --no-scaling > /dev/null 2>&1
done
python - <<'PY'
import csv,glob,collections,json
out={}
for f in sorted(glob.glob('gpurun_out/prof_r02/**/sq_*counter_collection.csv', recursive=True)):
    vals=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'mt_dense' in r['Kernel_Name'] and int(r['Grid_Size'])>1_000_000:
            vals[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in vals.items(): out[k]=sum(v)/len(v)
json.dump(out, open('gpurun_out/prof_r02/pmc_dense_sq.json','w'), indent=1)
print(json.dumps(out, indent=1))
PY
ls gpurun_out/prof_r02 | head -40

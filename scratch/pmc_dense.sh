cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc && mkdir -p gpurun_out/pmc
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc -o $tag -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-paths > /dev/null 2>&1
done
ls gpurun_out/pmc | head -20
python - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob('gpurun_out/pmc/*counter_collection.csv')):
    vals=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'mt_dense' in r['Kernel_Name'] and int(r['Grid_Size'])>10_000_000:
            vals[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in vals.items(): print(k, sum(v)/len(v), len(v))
PY

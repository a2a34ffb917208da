# SQ counters of the beam kernels at configs[3] (order 3): counters only, one pass per set
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_beam && mkdir -p gpurun_out/pmc_beam
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC" "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_beam -o $tag -- python scratch/cfg_beam.py ${1:-cfg4} > gpurun_out/pmc_beam/$tag.log 2>&1
done
python - <<'PY'
import csv,glob,collections
tot=collections.defaultdict(float)
for f in sorted(glob.glob('gpurun_out/pmc_beam/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'beam_' in k:
            tot[(k.split('(')[0][-44:], r['Counter_Name'])]+=float(r['Counter_Value'])
for k,v in sorted(tot.items()): print(k, f"{v:.4g}")
PY

# A/B of beam kernel variants on the GPU box: rebuild with DRT_EXTRA_FLAGS, profile the cfg step
# usage: VARIANTS="-DX=1|-DX=2" bash scratch/beam_lab.sh cfg4
cd $GRAFT_REPO_ROOT
IFS='|' read -ra VS <<< "${VARIANTS:-|-DBEAM_EXPAND_WAVES=5|-DBEAM_EXPAND_WAVES=6}"
for v in "${VS[@]}"; do
  echo "=== variant: [$v]"
  touch differt_amd/csrc/beam.hip
  DRT_EXTRA_FLAGS="$v" python -m differt_amd.build > /dev/null 2>&1 || echo BUILD FAILED
  bash scratch/prof_beam.sh ${1:-cfg4} 2>&1 | head -4
  grep -o '"s_per_step": [0-9.]*' gpurun_out/prof_beam_${1:-cfg4}/run.log | tail -1
  grep -o '"levels": [^]]*]' gpurun_out/prof_beam_${1:-cfg4}/run.log | tail -1
done
touch differt_amd/csrc/beam.hip; python -m differt_amd.build > /dev/null 2>&1

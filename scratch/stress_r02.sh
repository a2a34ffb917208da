# full-length randomized GPU-vs-oracle stress drivers, final round-2 code -> gpurun_out/stress_r02/*.json
mkdir -p gpurun_out/stress_r02
for s in oracle_stress query_oracle_stress trace_oracle_stress trace_stress hybrid_stress bvh_stress; do
  timeout 400 python scratch/$s.py 150 2> gpurun_out/stress_r02/$s.err | grep '^{' | tail -1 > gpurun_out/stress_r02/$s.json
  echo "$s: $(cat gpurun_out/stress_r02/$s.json | cut -c1-400)"
done
timeout 400 python scratch/beam_stress.py 150 2> gpurun_out/stress_r02/beam_stress.err | grep '^{' | tail -1 > gpurun_out/stress_r02/beam_stress.json
echo "beam_stress: $(cat gpurun_out/stress_r02/beam_stress.json | cut -c1-400)"

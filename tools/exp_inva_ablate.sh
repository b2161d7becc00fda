# tools/exp_inva_ablate.sh LIB : k_inv_a phase stamps at cfg1opt under the ablation bits of an experiment build
# (DSPSR_AMD_DEBUG: 1 no stores, 2 no loads, 4 no chirp loads, 256 part-major items; results wrong)
R=${GRAFT_REPO_ROOT:-$PWD}; mkdir -p $R/gpurun_out/r04u
for d in ${DBGS:-0 1 2 3 4 6 7 256}; do
  echo "== DSPSR_AMD_DEBUG=$d"
  DSPSR_AMD_DEBUG=$d DSPSR_AMD_LIB=$R/build/lib_$1.so timeout -k 10 120 python3 tools/stamps_cfg1.py > /tmp/inva_$d.txt 2>&1; grep -v amdgpu.ids /tmp/inva_$d.txt; grep -q "Memory access fault" /tmp/inva_$d.txt && exit 1
done

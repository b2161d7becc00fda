# tools/exp_inva_xpad.sh LIB : k_inv_a phase stamps at cfg1opt for paddings of the blocked spectrum's block stride (experiment build)
R=${GRAFT_REPO_ROOT:-$PWD}
for x in 0 32 96 160 544 2080 8224; do
  echo "== DSPSR_AMD_XPAD=$x"
  DSPSR_AMD_XPAD=$x DSPSR_AMD_LIB=$R/build/lib_$1.so timeout -k 10 120 python3 tools/stamps_cfg1.py 2>&1 | grep -v amdgpu.ids | head -4 || exit 1
done

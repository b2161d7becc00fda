# tools/exp_ablate.sh "ENV=VAL ..." lib1 lib2 ... : kernel stats of experiment builds under an ablation environment
# (e.g. DSPSR_AMD_DEBUG=3: no global loads and no stores in the passes -- results are wrong, the bench's parity gate ends
# the run after the timed region; the kernel trace is what is read)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
envs=$1; shift
for n in "$@"; do
  rm -rf gpurun_out/ab_$n
  env $envs DSPSR_AMD_LIB=$GRAFT_REPO_ROOT/build/lib_$n.so true
  ( export $envs DSPSR_AMD_LIB=$GRAFT_REPO_ROOT/build/lib_$n.so; timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$n -- python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-companions > gpurun_out/ab_$n.log 2>&1 )
  echo "== $n [$envs]"; python tools/kstats.py gpurun_out/ab_$n | grep -E "fwd_cols|fwd_rows|inv_chan<12, true"
done

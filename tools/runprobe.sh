cd $GRAFT_REPO_ROOT
for m in 0 1 2; do for run in 64 128 256 512 1024 4096 65536; do timeout -k 5 30 build/runprobe $m $run 131072 || exit 1; done; done

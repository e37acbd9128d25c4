cd ${GRAFT_REPO_ROOT:-/root/repo}; export CBGX_LIBRARY=$(pwd)/cbgbench_amd/lib/libcbgx_ablate.so
for a in 0 1024 2048 3072; do echo "abl=$a"; CBGX_BWD_ABL=$a bash scripts/gpu_train_stats.sh abl$a 12 2>&1 | grep "dgrad\|q_backward\|wgrad"; done

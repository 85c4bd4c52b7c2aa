# one-off: interleaved A/B of library builds given as arguments (step leg, then the rollout kernel at 65,536 / 131,072 envs)
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/ab_libs.sh "$@"
for round in 1 2 3; do for name in "$@"; do echo "== rollout round $round: $name"; G2048_LIB=build_ab/libg2048_$name.so timeout -k 10 120 python3 tools/rollout_rate.py 65536 131072 2>&1 | grep "envs:"; done; done

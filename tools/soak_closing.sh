cd ${GRAFT_REPO_ROOT:-/root/repo}
echo "## tools/oracle_full_games.py 2048 16"; timeout -k 10 400 python3 tools/oracle_full_games.py 2048 16 2>&1 | grep -v amdgpu.ids | tail -3
echo "## tools/soak_helpers.py 200"; timeout -k 10 300 python3 tools/soak_helpers.py 200 2>&1 | tail -1
echo "## tools/soak_history.py 100"; timeout -k 10 300 python3 tools/soak_history.py 100 2>&1 | tail -1
echo "## tools/soak_network.py 1024"; timeout -k 10 300 python3 tools/soak_network.py 1024 2>&1 | tail -1
echo "## smoke"; python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1

# long forms of the GPU tests on the final kernels (profiles/<round>/final_checks.txt)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
O=gpurun_out/final_checks; mkdir -p $O
{ echo "== fuzz_parity 160 cases, seed 6"; timeout 1500 python scripts/fuzz_parity.py 160 6 | tail -3
  echo "== determinism_check"; timeout 900 python scripts/determinism_check.py 2>&1 | tail -6
  echo "== stress_teams"; timeout 900 python scripts/stress_teams.py 2>&1 | tail -6
  echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3
  echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
} > $O/final_checks.txt 2>&1
cat $O/final_checks.txt

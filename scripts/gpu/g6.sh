set -u
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
O=gpurun_out/r06_g6; mkdir -p $O
for c in cfg1 cfg3 cfg4 cfg2; do timeout 600 python scripts/sched_ab.py $c "NMPC_TEAM_HELP=1" "NMPC_TEAM_HELP=2" "NMPC_TEAM_HELP=0" "NMPC_TEAM_HELP=1" "NMPC_TEAM_HELP=2"; done > $O/help_ab.jsonl 2> $O/help_ab.log; cat $O/help_ab.jsonl | cut -c1-220
for m in 1 2; do NMPC_TEAM_HELP=$m timeout 300 python scripts/latency_team.py cfg1 > $O/latency_help$m.json 2>> $O/lat.log; echo "latency help=$m: $(cut -c1-400 $O/latency_help$m.json)"; done

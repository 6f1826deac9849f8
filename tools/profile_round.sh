#!/bin/bash
# Produces the judged artefacts of a round on the GPU box: bench line, rocprofv3 kernel summary of the same command,
# and HBM traffic counters (separate --pmc passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes).
#   tools/profile_round.sh <tag> [bench.py arguments, e.g. --model conc]
R=$GRAFT_REPO_ROOT; TAG=${1:-r02}; shift; ARGS="$@"; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py $ARGS > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py $ARGS --no-cpu-baseline > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py $ARGS --no-cpu-baseline --no-roofline --steps 3 --warmup 1 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py $ARGS --no-cpu-baseline --no-roofline --steps 3 --warmup 1 > $O/pmc_write.log 2>&1
cat $O/bench.json

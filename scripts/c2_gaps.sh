# config 2 under rocprofv3 --kernel-trace: kernel durations and the idle time between consecutive launches (steady-state half of the run)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/c2tr
rocprofv3 --kernel-trace --output-format csv -d /tmp/c2tr -- python3 $R/scripts/c2_trace.py > /tmp/c2tr.log 2>&1
python3 $R/scripts/trace_gaps.py /tmp/c2tr

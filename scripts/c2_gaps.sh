# config 2 under rocprofv3 --kernel-trace: kernel durations and the idle time between consecutive launches (steady-state half of the run),
# then the launches around one restart
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/c2tr
rocprofv3 --kernel-trace --output-format csv -d /tmp/c2tr -- python3 $R/scripts/c2_trace.py > /tmp/c2tr.log 2>&1
python3 $R/scripts/trace_gaps.py /tmp/c2tr
echo "around a restart:"
python3 $R/scripts/trace_window.py /tmp/c2tr k_panel_mult

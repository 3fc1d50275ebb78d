# Round 5: schedule knobs of PoseEngine.submit on one box (tools/step_times.py, 60 steps, two repetitions each, interleaved
# with the default so that box drift shows): stream priorities, stream counts, where the stage merge runs.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5h; mkdir -p $O; export PYTHONUNBUFFERED=1
run() { name=$1; shift; for rep in 1 2; do env "$@" timeout 120 python tools/step_times.py --steps 60 --warmup 10 2>/dev/null | grep "ms/step" | sed "s/^/$name rep$rep: /"; done; }
{
run default LP_NET_PRIO=0
run net_hi LP_NET_PRIO=-1
run net_hi_net3 LP_NET_PRIO=-1 LP_NET_STREAMS=3
run net_hi_early LP_NET_PRIO=-1 LP_SPLIT=early
run net_hi_lanes6 LP_NET_PRIO=-1 LP_LANES=6
run net_hi_ae2 LP_NET_PRIO=-1 LP_AE_STREAMS=2
run default_again LP_NET_PRIO=0
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')"
} > $O/sched_experiments.txt 2>&1
cat $O/sched_experiments.txt

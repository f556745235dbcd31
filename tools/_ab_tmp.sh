cd $GRAFT_REPO_ROOT
for i in 1 2; do
python tools/ab_pipeline.py --mode single 2>/dev/null | tail -1
python tools/ab_pipeline.py --mode dual 2>/dev/null | tail -1
python tools/ab_pipeline.py --mode dual --prio 2>/dev/null | tail -1
python tools/ab_pipeline.py --mode quad 2>/dev/null | tail -1
done

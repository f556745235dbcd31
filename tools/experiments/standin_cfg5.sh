cd $GRAFT_REPO_ROOT
for shape in none real 512x157286x151x35 512x124500x151x35 512x73400x151x35 512x38600x151x35 512x38600x80x35; do
python tools/experiments/ab_light_update_hot.py --scenes 16 --goals 64 --waypoints 50 --objects 12 --shape $shape 2>&1 | tail -1
done

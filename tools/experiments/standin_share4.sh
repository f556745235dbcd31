cd $GRAFT_REPO_ROOT
for shape in none real 512x96256x151x21 512x66000x124x21 512x31744x80x21 512x31744x80x30 256x31744x88x30 512x96256x151x12 512x31744x80x12; do
python tools/experiments/ab_light_update_hot.py --scenes 13 --goals 128 --shape $shape 2>&1 | tail -1
done
for shape in none real 512x96256x151x21 512x31744x80x21; do
python tools/experiments/ab_light_update_hot.py --scenes 25 --goals 64 --shape $shape 2>&1 | tail -1
done

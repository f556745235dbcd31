cd $GRAFT_REPO_ROOT
OMGX_GS_WIDE=0 python tools/gs_phase_clock.py 100 64 50 > gpurun_out/r06g_clock_w4_n50.json 2> gpurun_out/r06g_clock_w4_n50.err
OMGX_GS_WIDE6_LONG_MAX=100000000 python tools/gs_phase_clock.py 100 64 50 > gpurun_out/r06g_clock_w6_n50.json 2> gpurun_out/r06g_clock_w6_n50.err
python - <<P
import json
for t in ("w4","w6"):
    try:
        d=json.load(open("gpurun_out/r06g_clock_%s_n50.json"%t))["timeline_us"]
        print(t, "span", d["kernel_span"], "life", d["goal_wg_duration_mean/p50/p90/max"], "occ", d["slot_occupancy_by_tenth"])
    except Exception as e:
        print(t, "ERR", e, open("gpurun_out/r06g_clock_%s_n50.err"%t).read()[-600:])
P

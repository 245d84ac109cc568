# in-kernel stage ticks (prebuilt -DSS_PROFILE variant), headline workload; SELFCOL=1 for the body-body contact path
for sc in ${SELFCOLS:-0 1}; do
SS_PROF_LIB=$PWD/smplsim_amd/variants/libsmplsim_hip_prof.so SELFCOL=$sc NENV=${NENV:-4096} STEPS=${STEPS:-30} python tools/stage_profile.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('selfcol $sc mean iters',round(d['mean_newton_iters'],1)); print({k:round(v['ticks_per_mj_step_per_wave']) for k,v in d['stages'].items() if v['ticks_per_mj_step_per_wave']>0.5 and not k.startswith('n:')}); print({k:round(v['ticks_per_mj_step_per_wave'],4) for k,v in d['stages'].items() if k.startswith('n:')})"
done

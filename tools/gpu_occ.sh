# per-wave stage ticks versus resident waves per CU (SS_ENVS_PER_WG caps the workgroup size)
for e in 1 4 8 12; do
  SS_ENVS_PER_WG=$e NENV=$((e*256*2)) STEPS=20 python tools/stage_profile.py > /dev/null 2>gpurun_out/stage.err
  python - <<PY
import json
d=json.load(open('gpurun_out/stage_profile.json'))
s=d['stages']
print('envs/CU=$e', ' '.join('%s=%.0f'%(k[:10],v['ticks_per_mj_step_per_wave']) for k,v in s.items() if v['ticks_per_mj_step_per_wave']>0))
PY
done

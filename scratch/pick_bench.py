import json,sys
for f in sys.argv[1:]:
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    r=d['extras'].get('closed_loop_rollout',{})
    print(f, 'value',round(d['value']), 'ms',round(d['ms_per_step'],1),'conv_last',d['solved_frac_last_tick'],'| rollout',round(r.get('converged_solves_per_s',0)), r.get('converged_frac'), r.get('passes_per_instance'), r.get('ip_iterations_launched'))

# same-box A/B of the step against a few switches (gpurun -- bash profiles/micro/knob_sweep_r04.sh); 60 timed + 150 steady steps each
run() { env "$@" python bench.py --steps 60 --no-cpu-baseline --no-parity --no-secondary --no-profile 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); g=j['teacher_gate']; print('$*', j['ms_per_step'], j['ms_per_step_steady'], 'gate opened', g['opened'], 'timeouts', g['timeouts'])"; }
run A=1
run MAGIC_TEACHER_GATE_US=0
run MAGIC_TEACHER_GATE_US=200
run MAGIC_TEACHER_GATE_US=800
run MAGIC_TEACHER_GATE_RECENT_US=50
run MAGIC_DW_ATOMICS=1
run MAGIC_DW_GROUP=48
run MAGIC_RBW_ROWS=32
run A=2

export TMPDIR=/tmp
df -h /tmp /dev/shm | tail -2; free -g | head -2; nproc
timeout 900 python -m pytest tests/test_group_kernel.py tests/test_cboc.py -x -q -m gpu 2>&1 | tail -3
fmt='import json,sys
for ln in sys.stdin:
    if ln.startswith("{"):
        d=json.loads(ln); r=d["roofline"]; print(sys.argv[1], "value", d["value"], "ms/step", d["ms_per_step"], "solo_kernel_ms", r["avg_kernel_ms"], "walk", r["avg_walk_ms"], "chk", d["config"]["output_checksum"], "fam", d["config"]["kernel_family"])'
for a in "" "--pipeline 1" "--signal cboc" "--signal cboc --channels 9" "--channels 9"; do python bench.py --no-extras --no-cpu-baseline $a 2>/dev/null | python -c "$fmt" "[$a]"; done
timeout 900 tools/config5_sites.sh
timeout 1500 python -m pytest tests/test_cli.py -x -q -m gpu -k "config5" 2>&1 | tail -5

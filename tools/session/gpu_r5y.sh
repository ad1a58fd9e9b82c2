#!/bin/bash
# round 5, session y: fp32 Paul's last two two-pass rows as overlap-save rows on blocks of four tiles (ols_big = 2)
export TMPDIR=/tmp
OUT=gpurun_out/r5y; mkdir -p $OUT
run() {  # label config opts...
  local label=$1 c=$2; shift 2
  timeout 300 python bench.py --config $c --steps 40 --warmup 5 --no-cpu-baseline --no-extra --no-live-traffic "$@" --detail $OUT/${label}_$c.json > $OUT/${label}_$c.line 2> $OUT/${label}_$c.err
  python - <<P
import json
d=json.load(open("$OUT/${label}_$c.json"))
pc=d["roofline"]["per_class"]
par=d.get("parity") or {}
print("$label $c: %.4f ms  %.1f GS/s  " % (d["ms_per_step"], d["value"]) + "  ".join("%s %d rows %.2f us/row" % (k, v["rows"], v["us_per_row"]) for k, v in pc.items()), " parity", par.get("max_row_err"), par.get("ok"))
P
}
for rep in 1 2; do
  run base$rep c3_paul
  run big2_$rep c3_paul --opt ols_big=2
  run big2h_$rep c3_paul --opt ols_big=2 --opt ols_big4_min_halo=4097
done
run base c3_dog
run big2h c3_dog --opt ols_big=2 --opt ols_big4_min_halo=4097
run base c2
run big2h c2 --opt ols_big=2 --opt ols_big4_min_halo=4097
run base paul64
run big2h paul64 --opt ols_big=2 --opt ols_big4_min_halo=4097
# every comparable row of that variant against the oracle (the bench's own parity block)
timeout 600 python bench.py --config c3_paul --steps 10 --warmup 3 --no-extra --no-live-traffic --opt ols_big=2 --opt ols_big4_min_halo=4097 --detail $OUT/parity_c3_paul.json > $OUT/parity_c3_paul.line 2> $OUT/parity_c3_paul.err
python -c "import json; d=json.load(open('$OUT/parity_c3_paul.json')); print('parity', d.get('parity'))"
echo done

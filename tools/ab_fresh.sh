# same-box A/B of fresh-points steps: lib/libhgwarp_prev.so against lib/libhgwarp.so, bench.py --points both, alternating
for rep in 1 2; do for lib in prev cur; do
  if [ $lib = prev ]; then export HGWARP_LIB=$PWD/homography.js_amd/lib/libhgwarp_prev.so; else unset HGWARP_LIB; fi
  for c in C2 C3 C4 C5; do e=""; [ $c = C5 ] && e="--frames 8"; python bench.py --config $c $e --points both --sources shared --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); x=d.get('roofline_fresh',{}); print('$lib', '$c', 'resident', d['ms_per_step'], 'fresh', x.get('ms_per_step'), 'ratio', round(x.get('ms_per_step',0)/d['ms_per_step'],3), d.get('verified'))
"; done; done; done

"""Annotate a PMC summary of the map kernel (scripts/pmc_summary.py output) with the derived figures
bench.py and DESIGN.md quote, and write it where bench.py looks for it.
    python3 scripts/pmc_finish.py gpurun_out/pmc_<tag>.json profiles/r04_pmc_map.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import map_source_hash   # noqa: E402

src, dst = sys.argv[1], sys.argv[2]
d = json.load(open(src))
per = d['per_launch']
d['derived']['l2_misses_per_pair'] = per['TCC_MISS_sum'] / 1e7
d['derived']['note'] = (
    'FETCH_SIZE*1024 = %.1f GB agrees with TCC_MISS*64 B = %.1f GB (random 16-B probes: one 64-B sector per miss; '
    'Infinity-Cache hits are counted in FETCH_SIZE); L2 miss rate against the measured random-gather ceiling of 52 G/s '
    '(profiles/r01_gather_ceiling.log)' % (per['FETCH_SIZE'] * 1024 / 1e9, d['derived']['tcc_miss_bytes_at_64B'] / 1e9))
d['source_hash'] = map_source_hash()     # bench.py quotes roofline.traffic only from a summary of ITS kernel sources
d['command'] = ('rocprofv3 --kernel-trace --pmc <one set per pass> --output-format csv -- python3 scripts/profile_map.py '
                '--reps 2 --cache /tmp/skm_idx.npz   (scripts/pmc_map.sh via scripts/round_profiles.sh; four separate '
                'passes, summarised by scripts/pmc_summary.py)')
d['workload'] = ('configs[1] stand-in: T190k index (2 GiB k-mer table + 4 GiB bucket copy), 10 M 2x100 pairs, one '
                 'map_units_kernel<false, true> launch')
json.dump(d, open(dst, 'w'), indent=1)

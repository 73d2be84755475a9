set -e
for pad in 0 3000 8000 15000 35000; do echo "pad $pad"; X265AMD_LDS_PAD=$pad timeout -k 10 120 python tools/search_bench.py --reps 5 2>&1 | grep -v "^$" ; done

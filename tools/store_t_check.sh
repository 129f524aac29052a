bash tools/ab_env.sh DGP_STORE_T 0 1 12496 | tail -4
for v in 0 1; do DGP_STORE_T=$v timeout -k 10 300 python tools/configs_check.py 2>&1 | grep -E "config 4|config 2-alt|config 2 \(" | sed "s/^/STORE_T=$v /"; done

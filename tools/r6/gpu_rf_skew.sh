cd "$GRAFT_REPO_ROOT"; EV2G_REFILL_STAMPS=1 timeout 200 python tools/refill_time.py cfg2 2>&1 | grep -E "refill:|stamps|refill of" | tail -4 | cut -c1-260

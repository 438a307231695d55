for q in 2 3 4 6 8 16; do for h in 0 1; do
echo -n "queues=$q serial_heads=$h: "; GPU_MAX_HW_QUEUES=$q RVA_SERIAL_HEADS=$h RVA_TUNE_IN_PLAN=0 python tools/two_streams.py 2>&1 | grep "streams\|stream" | tr '\n' ' '; echo
done; done

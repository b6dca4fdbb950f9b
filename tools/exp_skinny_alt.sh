for a in 0 256 1 2 4 8 16 32 64 128 0; do echo "ALT=$a"; SL_SKINNY_ALT=$a python tools/time_decode_step.py 1 8 2>&1 | grep "B="; done

for a in ${ABLS:-0 32 33}; do echo "ABL=$a"; RB_V2_ABL=$a timeout 300 python tools/prof_dvis.py variants 32 f16x3-v2 2>&1 | tail -1; done

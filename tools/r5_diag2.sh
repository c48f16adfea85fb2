#!/bin/bash
out=gpurun_out/r5_diag3.txt
: > $out
python tools/r5_diag2.py retained=off dissection=off >> $out 2>&1
python tools/r5_diag2.py retained=off dissection=auto timing=1 >> $out 2>&1
python tools/r5_diag2.py retained=on dissection=off timing=1 >> $out 2>&1
python tools/r5_diag2.py retained=on dissection=auto timing=1 >> $out 2>&1
SK_BS_RESIDENT=0 python tools/r5_diag2.py retained=on dissection=auto timing=1 >> $out 2>&1
cat $out

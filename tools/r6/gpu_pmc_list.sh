#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/$1; mkdir -p $O
rocprofv3 -L > $O/counters_raw.txt 2>&1
grep -oE "(Name|name)[: ]+[A-Za-z0-9_]+" $O/counters_raw.txt | awk '{print $NF}' | sort -u > $O/counter_names.txt
wc -l $O/counter_names.txt
grep -E "^(TCP|TA_|TD_|TCC_|SQ_INST|SQ_WAIT|SQ_ACTIVE|SQ_LDS|SQ_BUSY|SQC|GRBM|SPI|TCA)" $O/counter_names.txt | tr '\n' ' ' | fold -w 200

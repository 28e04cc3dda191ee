#!/bin/bash
# HIP-graph runtime knobs vs the graphed sampler (ms per 20-NFE pass)
run() { echo -n "$1 => "; env $1 python bench.py --mode sample --no-cpu-baseline --no-roofline --big-batch 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), round(d['value']))" 2>&1 | tail -1; }
run "SDMI_SAMPLE_SPLIT=1"
run "SDMI_SAMPLE_SPLIT=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"
run "SDMI_SAMPLE_SPLIT=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1"
run "SDMI_SAMPLE_SPLIT=2 DEBUG_HIP_FORCE_GRAPH_QUEUES=2"
run "SDMI_SAMPLE_SPLIT=2 DEBUG_HIP_FORCE_GRAPH_QUEUES=4"
run "SDMI_SAMPLE_SPLIT=4 DEBUG_HIP_FORCE_GRAPH_QUEUES=4"
run "SDMI_SAMPLE_SPLIT=2 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"
run "SDMI_SAMPLE_SPLIT=1 DEBUG_HIP_GRAPH_BATCH_SIZE=1024"

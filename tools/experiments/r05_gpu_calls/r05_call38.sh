#!/bin/bash
mkdir -p gpurun_out
{
echo "== knob sweep 2 (hip = always sleep 1)"; timeout 600 bash tools/ab_run.sh hip nobs sl2 nobsnors nobsnogrp nobsnop128 nobssl2
} > gpurun_out/r05_call38.log 2>&1
cat gpurun_out/r05_call38.log | tail -70

#!/bin/bash
for rep in 1 2; do for l in hip m_gemm2 m_gemm m_highway m_prenet m_gemm2ilp m_gemmilp; do echo -n "$l: "; TACO_LIB=$PWD/tacotron_amd/libtaco_$l.so timeout 200 python tools/family_trace.py 2>&1 | grep -E "^step|^sum" | tr '\n' ' '; echo; done; done

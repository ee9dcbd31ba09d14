#!/bin/bash
# ONE parametrised GPU trip runner (round 6; replaces the per-experiment gpu_*.sh wrappers of rounds 2-5).
#   gpurun --timeout S -- 'bash tools/gpu_trip.sh TAG step[:arg[:arg..]] ...'
# Every step writes under gpurun_out/TAG/ and appends one line to gpurun_out/TAG/summary.txt.  Steps:
#   pytest:<expr>            pytest -m gpu on the files / -k expression given (':'-separated words become arguments)
#   suite                    the whole GPU suite
#   smoke                    __graft_entry__.smoke()
#   p9ab:<tiles>:<shapes>    tools/p9_ab.py on diag tiles / shapes (comma lists)
#   c5ab:<VAR>:<v1,v2,..>:<reps>[:extra bench flags]   configs[4] bench alternating over the values of one NOMAD_* switch (diag library)
#   c2ab:<VAR>:<v1,v2,..>:<reps>[:extra bench flags]   the same for the fp32 headline ('+' in the extra flags stands for a space: --dtype+bf16x3)
#   c4ab:<VAR>:<v1,v2,..>:<reps>                       the same for configs[3] (tools/bench_c4.py)
#   bench                    default bench.py (the driver's line) -> bench.json
#   c5table:<name>[:VAR=val,..]   rocprofv3 kernel trace of the single-stream configs[4] bench -> layer_table_<name>.json + kernel_stats_<name>.csv
#   c2stats:<name>           rocprofv3 --kernel-trace --stats of bench.py --single-stream -> kernel_stats_<name>.csv
#   c4table:<name>[:VAR=val,..]   kernel trace of configs[3] -> c4_trace_table_<name>.txt
#   pmc:<shape>:<tile>:<name>     PMC passes (matrix pipe, TA, waits) of one bf16 GEMM shape
#   pmcpy:<script>:<kernel substr>:<name>   the same PMC passes (+ SALU / waves / LDS conflicts) over any tools/ script
#   attnab:<VAR>:<v1,v2,..>  tools/attn_bf16_ab.py per value
#   attnf32:<VAR>:<v1,v2,..> tools/attn_f32_time.py per value
#   py:<script>[:args]       python3 tools/<script> args  (stdout -> <script>.log)
TAG=${1:-trip}; shift
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
SUM=$OUT/summary.txt
C5FLAGS="--dtype bf16 --seconds 30 --batch 32 --refs 4 --steps 10 --warmup 3 --no-cpu-baseline --no-profile"

val() { python3 -c "import json,sys; d=json.load(open('$1')); print(' '.join(str(d.get(k)) for k in sys.argv[1:]))" "${@:2}" 2>/dev/null; }

for step in "$@"; do
  IFS=':' read -r -a S <<< "$step"
  case ${S[0]} in
    pytest)
      ARGS=(); for a in "${S[@]:1}"; do ARGS+=("${a//+/ }"); done   # '+' inside an argument stands for a space (-k:"a+or+b")
      timeout 2400 python -m pytest "${ARGS[@]}" -q -m gpu -x --timeout 1200 > $OUT/pytest_$(echo "${S[1]}" | tr '/ ' '__').log 2>&1
      echo "pytest ${S[*]:1}: exit $? $(tail -n 1 $OUT/pytest_$(echo "${S[1]}" | tr '/ ' '__').log)" | tee -a $SUM ;;
    suite)
      timeout 3000 python -m pytest tests -q -m gpu -x --timeout 1200 > $OUT/suite.log 2>&1
      echo "suite: exit $? $(tail -n 1 $OUT/suite.log)" | tee -a $SUM ;;
    smoke)
      timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke: exit $? $(tail -n 2 $OUT/smoke.log | tr '\n' ' ')" | tee -a $SUM ;;
    p9ab)
      timeout 900 python3 tools/p9_ab.py --tiles ${S[1]} --shapes ${S[2]} ${S[3]:+--vendor ${S[3]}} > $OUT/p9_ab_${S[1]//,/_}.jsonl 2> $OUT/p9_ab.err
      echo "p9ab exit $?" | tee -a $SUM; cat $OUT/p9_ab_${S[1]//,/_}.jsonl | tee -a $SUM ;;
    c5ab|c2ab|c4ab)
      VAR=${S[1]}; IFS=',' read -r -a VALS <<< "${S[2]}"; REPS=${S[3]:-2}
      for rep in $(seq 1 $REPS); do for v in "${VALS[@]}"; do
        f=$OUT/${S[0]}_${VAR}_${v}_$rep.json
        if [ ${S[0]} = c5ab ]; then env NOMAD_DIAG_LIB=1 $VAR=$v timeout 600 python bench.py $C5FLAGS ${S[4]//+/ } > $f 2> ${f%.json}.err
          echo "c5 $VAR=$v rep $rep: $(val $f value ms_per_step)" | tee -a $SUM
        elif [ ${S[0]} = c2ab ]; then env NOMAD_DIAG_LIB=1 $VAR=$v timeout 600 python bench.py --no-cpu-baseline --no-profile --no-also --live-traffic off ${S[4]//+/ } > $f 2> ${f%.json}.err
          echo "c2 $VAR=$v rep $rep: $(val $f value ms_per_step)" | tee -a $SUM
        else env NOMAD_DIAG_LIB=1 $VAR=$v timeout 600 python3 tools/bench_c4.py > $f 2> ${f%.json}.err
          echo "c4 $VAR=$v rep $rep: $(tail -n 1 $f)" | tee -a $SUM
        fi
      done; done ;;
    bench)
      timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?: $(val $OUT/bench.json value ms_per_step)" | tee -a $SUM ;;
    c5table|c4table|c2stats)
      NAME=${S[1]:-x}; ENVS=$(echo "${S[2]}" | tr ',' ' ')
      if [ ${S[0]} = c5table ]; then CMD="$ROOTDIR/bench.py $C5FLAGS --steps 5 --warmup 2 --single-stream"
      elif [ ${S[0]} = c2stats ]; then CMD="$ROOTDIR/bench.py --no-cpu-baseline --no-profile --no-also --live-traffic off --single-stream --steps 4 --warmup 2"
      else CMD="$ROOTDIR/tools/c4_profile.py 20"; fi
      (cd /tmp && env NOMAD_DIAG_LIB=1 $ENVS timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$NAME -o t -- python3 $CMD > $OUT/prof_$NAME.json 2> $OUT/prof_$NAME.err)
      echo "${S[0]} $NAME exit $?" | tee -a $SUM
      f=$(find $OUT/prof_$NAME -name "*kernel_trace.csv" | head -1); g=$(find $OUT/prof_$NAME -name "*kernel_stats.csv" | head -1)
      [ -n "$g" ] && cp $g $OUT/kernel_stats_$NAME.csv
      if [ -n "$f" ]; then
        if [ ${S[0]} = c5table ]; then python3 tools/c5_layer_table.py $f > $OUT/layer_table_$NAME.json && cat $OUT/layer_table_$NAME.json | tee -a $SUM
        elif [ ${S[0]} = c4table ]; then python3 tools/c4_trace_table.py $f > $OUT/c4_trace_table_$NAME.txt 2>&1; head -n 40 $OUT/c4_trace_table_$NAME.txt; fi
      fi
      rm -rf $OUT/prof_$NAME ;;
    pmc)
      SHAPE=${S[1]}; TILE=${S[2]}; NAME=${S[3]:-$SHAPE}
      i=0
      for grp in "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_VMEM" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"; do
        i=$((i+1))
        (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_${NAME}_p$i -o p -- python3 $ROOTDIR/tools/gemm_bf16_one.py $SHAPE $TILE > $OUT/pmc_${NAME}_p$i.log 2>&1)
      done
      python3 tools/pmc_summary.py $OUT "pmc_${NAME}_p" gemm_bf16 > $OUT/pmc_$NAME.txt; cat $OUT/pmc_$NAME.txt | tee -a $SUM
      find $OUT -name "*.csv" -path "*pmc_${NAME}_p*" -delete ;;
    pmcpy)   # pmcpy:<tools script>:<kernel name substring>:<name>[:script args] - the same PMC passes over any tools/ script
      SCRIPT=${S[1]}; SUB=${S[2]}; NAME=${S[3]:-$SCRIPT}
      i=0
      for grp in "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_VMEM" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_SALU SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
        i=$((i+1))
        (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_${NAME}_p$i -o p -- python3 $ROOTDIR/tools/$SCRIPT ${S[@]:4} > $OUT/pmc_${NAME}_p$i.log 2>&1)
      done
      python3 tools/pmc_summary.py $OUT "pmc_${NAME}_p" "$SUB" > $OUT/pmc_$NAME.txt; cat $OUT/pmc_$NAME.txt | tee -a $SUM
      find $OUT -name "*.csv" -path "*pmc_${NAME}_p*" -delete ;;
    attnab)
      VAR=${S[1]}; IFS=',' read -r -a VALS <<< "${S[2]}"
      for rep in 1 2; do for v in "${VALS[@]}"; do env NOMAD_DIAG_LIB=1 $VAR=$v timeout 300 python3 tools/attn_bf16_ab.py >> $OUT/attn_ab.jsonl 2>> $OUT/attn_ab.err; done; done
      cat $OUT/attn_ab.jsonl | tee -a $SUM ;;
    attnf32)
      VAR=${S[1]}; IFS=',' read -r -a VALS <<< "${S[2]}"
      for rep in 1 2; do for v in "${VALS[@]}"; do env NOMAD_DIAG_LIB=1 $VAR=$v timeout 300 python3 tools/attn_f32_time.py >> $OUT/attn_f32.jsonl 2>> $OUT/attn_f32.err; done; done
      cat $OUT/attn_f32.jsonl | tee -a $SUM ;;
    py)
      timeout 1200 python3 tools/${S[1]} ${S[@]:2} > $OUT/$(basename ${S[1]} .py).log 2> $OUT/$(basename ${S[1]} .py).err; echo "py ${S[*]:1}: exit $?" | tee -a $SUM
      tail -n 30 $OUT/$(basename ${S[1]} .py).log ;;
    *) echo "unknown step ${S[0]}" | tee -a $SUM ;;
  esac
done

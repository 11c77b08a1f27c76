#!/bin/bash
cd $GRAFT_REPO_ROOT
fail=0
for loss in cosine featdist kd barlow; do for opt in rmsprop adamw adam lars; do for dt in bf16 f32; do
  out=$(timeout -k 10 120 python LstmDistillFromDinoV2Train.py --synthetic 96 --batch_size 16 --num_epochs 2 --validation_frequency 1 --log_dir gpurun_out/m_$loss$opt$dt --hidden_size 128 --lstm_layers 2 --loss $loss --optimizer $opt --dtype $dt 2>&1); rc=$?
  last=$(echo "$out" | grep "EPOCH 1" | tail -1 | cut -c1-90)
  echo "$loss $opt $dt rc=$rc $last"
  if [ $rc -ne 0 ]; then fail=1; echo "$out" | grep -v Warn | tail -5; fi
done; done; done
for args in "--synthetic 96 --batch_size 16 --num_epochs 2 --validation_frequency 1 --hidden_size 128"; do
  out=$(timeout -k 10 200 python LstmDistillFromDinoV2TrainSpampinato.py $args --log_dir gpurun_out/m_sp 2>&1); rc=$?; echo "spampinato rc=$rc $(echo "$out" | grep 'EPOCH 1' | tail -1 | cut -c1-90)"; [ $rc -ne 0 ] && { fail=1; echo "$out" | grep -v Warn | tail -5; }
done
echo "matrix fail=$fail"

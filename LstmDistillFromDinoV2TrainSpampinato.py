#!/usr/bin/env python3
"""MI355X trainer with the command-line / checkpoint surface of the reference's
LstmDistillFromDinoV2TrainSpampinato.py (BASELINE.json configs[3]: the Spampinato split, 128 channels x 440 samples).

    python LstmDistillFromDinoV2TrainSpampinato.py --eeg_dataset ... --eeg_dataset_split ... --teacher_features f.npy
    python LstmDistillFromDinoV2TrainSpampinato.py --synthetic 256 --time_high 460 --hidden_size 1024 --lstm_layers 2

Same loop as LstmDistillFromDinoV2Train.py (one implementation, `Flavour` picks what differs), with this script's
choices (reference file:line): AdamW at 1e-4 (:194,378), 200 epochs (:198), `Model(input_size=128, lstm_size=128,
lstm_layers=4, output_size=D, include_top=False)` (:368), loss_fn_kd with alpha / temperature from --hyperprams
(default alpha 0: pure cross-entropy of the logits against the class id, :107-121,262,288-296), weights-only resume
from --custom_model_weights when that file exists (:369-371), the first validated best saved as
lstm_dinov2_epoch_{EPOCH}_best_loss.pth and every later improvement as lstm_dinov2_epochs_{EPOCHS}_best_loss.pth
(:467-475), the Spampinato dataset reader (subject filter + split file, per-channel (x - means) / stddevs).
"""
import LstmDistillFromDinoV2Train as _loop


def main(argv=None):
    return _loop.main(argv, flavour=_loop.SPAMPINATO)


if __name__ == "__main__":
    main()

set -e
mkdir -p /tmp/tcli && cd /tmp/tcli
cat > img_small.json <<'J'
{"vocab_size": 28996, "hidden_size": 64, "num_hidden_layers": 2, "num_attention_heads": 4, "intermediate_size": 128, "max_position_embeddings": 512, "type_vocab_size": 2, "hidden_act": "gelu", "hidden_dropout_prob": 0.1, "attention_probs_dropout_prob": 0.1, "initializer_range": 0.02}
J
cat > train.json <<'J'
{"txt_model_config": "bert-base-cased", "img_model_config": "/tmp/tcli/img_small.json", "itm_global_file": null, "seed": 42, "output_dir": "/tmp/tcli/out", "max_txt_len": 60, "conf_th": 0.2, "max_bb": 100, "min_bb": 10, "num_bb": 36, "project_dim": 64, "train_batch_size": 16, "valid_batch_size": 32, "num_train_epochs": 2, "learning_rate": 0.001, "num_hard_negatives": 2, "sample_init_hard_negatives": true, "hard_negatives_sampling": "hard", "fp16": false}
J
cd $GRAFT_REPO_ROOT && python -m lightningdot_amd.train_itm --config /tmp/tcli/train.json --synthetic 64 2>&1 | tail -4; ls /tmp/tcli/out

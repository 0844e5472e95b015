#!/bin/bash
# OPA-DPO training launcher (MI355X): one process per GPU, RCCL over xGMI.  Same knobs as the reference's
# run/train_opa_dpo.sh; override through the environment, e.g.  GPUS_PER_NODE=8 bash run/train_opa_dpo.sh
set -e
export HSA_ENABLE_IPC_MODE_LEGACY=0
export GPUS_PER_NODE=${GPUS_PER_NODE:-8}
export IMAGE_DIR=${IMAGE_DIR:-none}
BASE_MODEL=${BASE_MODEL:-./base_models/llava-v1.5-7b}
POLICY_LORA=${POLICY_LORA:-./output/llava7b_opa_model/checkpoint-final}
DATA=${DATA:-./base_datasets/opadpo_training_data-7B}
OUT=${OUT:-./output/llava7b_opadpo_model}
python opa-dpo_amd/build.py
torchrun --standalone --nnodes=1 --nproc-per-node=$GPUS_PER_NODE --local-addr 127.0.0.1 \
    opadpo/opadpo_train_custom.py \
    --cfg configs/llava/llava_dpo.yaml \
    --base_model_name $BASE_MODEL --policy_model_name_or_path $POLICY_LORA --data_path $DATA --output_dir $OUT \
    --lora_rank 256 --lora_alpha 512 --lora_drop 0.0 \
    --rollout_batch_size 64 --step_batch_size 32 --rollout_per_device_batch_size 2 --step_per_device_batch_size 2 \
    --reward_model_per_device_batch_size 2 --noptepochs 1 --total_epochs 4 --max_step 300 --save_steps 75 \
    --learning_rate 1e-6 --warmup_steps 5 --max_grad_norm 1.0 --weight_decay 0.0 \
    --query_len 128 --response_len 896 --model_max_length 2048 --temperature 1.0 \
    --beta 0.1 --CoPO True --CoPO_method random --CoPO_mask_ratio 0.3 --CoPO_coef 0.2 --mDPO_anchor True --Anchor_value 0.0 \
    --detailed_report True --response_score True --response_image_relation True \
    --image_aspect_ratio pad --ddp_backend nccl --report_to none "$@"

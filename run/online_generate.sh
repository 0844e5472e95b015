#!/bin/bash
# Rollout generation launcher (MI355X): one process per GPU, prompts strided by rank, no collective.  Same knobs as the reference's
# run/online_generate.sh; the GPT-4V feedback call of that stage needs the network and is not part of this build - the JSON
# files carry the sampled responses with empty feedback fields.
set -e
export HSA_ENABLE_IPC_MODE_LEGACY=0
export GPUS_PER_NODE=${GPUS_PER_NODE:-8}
DATA_DIR=${DATA_DIR:-./base_datasets/LLaVA-RLAIF-SubData/subset1}
MODEL_DIR=${MODEL_DIR:-./base_models/llava-v1.5-7b}
POLICY_LORA_DIR=${POLICY_LORA_DIR:-none}
OUTPUT_DIR=${OUTPUT_DIR:-./output/llava7b_online_generation_subset1}
python opa-dpo_amd/build.py
torchrun --standalone --nnodes=1 --nproc-per-node=$GPUS_PER_NODE --local-addr 127.0.0.1 \
    opadpo/online_generation_custom.py \
    --cfg configs/llava/llava_online_generation.yaml \
    --base_model_name $MODEL_DIR --base_model $MODEL_DIR --policy_model_name_or_path $POLICY_LORA_DIR \
    --output_dir $OUTPUT_DIR --image_folder $DATA_DIR --data_path $DATA_DIR --seed 42 \
    --rollout_batch_size 32 --step_batch_size 32 --rollout_per_device_batch_size 4 --reward_model_per_device_batch_size 4 \
    --step_per_device_batch_size 4 --total_epochs 1 --model_max_length 2048 --query_len 128 --response_len 896 --noptepochs 1 \
    --mm_vision_select_layer -2 --ddp_backend nccl --top_k 30 --top_p 0.95 --temperature 1.0 --phase 0 --sample_num 2500 "$@"

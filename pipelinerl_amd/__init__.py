"""pipelinerl_amd — MI355X-native implementation of PipelineRL's rollout -> preprocess -> finetune
hot path.  The package mirrors the reference's module names for the path:

    pipelinerl_amd.finetune.rl      rl_step, populate_rl_data, RLConfig      (HIP kernels K1-K5)
    pipelinerl_amd.finetune.data    collate, collate_packed, preprocess_fn   (HIP kernels K6/K7)
    pipelinerl_amd.finetune.types   PipelineBatchEncoding
    pipelinerl_amd.streams          read_stream / write_to_streams
    pipelinerl_amd.shared_memory_array  SharedMemoryQueue
    pipelinerl_amd.finetune_loop    LearnerStep.step(), WeightUpdateManager
    pipelinerl_amd.rollouts         TrainingText, RolloutResult (plugin return types)

libprl.so (C ABI in include/prl.h) is loaded lazily by `pipelinerl_amd._lib.load()`; importing the
package itself does not need a GPU.
"""

__version__ = "0.1.0"

class PipelineCallback:
    tensor_inputs = []


class MultiPipelineCallbacks:
    tensor_inputs = []

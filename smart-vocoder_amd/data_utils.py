"""Import-compatibility module: the reference's inference notebook imports ``AudioSpecLoader`` and
``AudioSpecCollate`` from data_utils (inference.ipynb cell 0) without using them.  The training data pipeline
(reference data_utils.py) is out of scope; the names exist so the notebook's imports succeed, and raise on use."""


class _TrainingOnly:
    def __init__(self, *a, **k):
        raise NotImplementedError(f"{type(self).__name__} belongs to the reference's training data pipeline, "
                                  "which the MI355X inference path does not include")


class AudioSpecLoader(_TrainingOnly):
    pass


class AudioSpecCollate(_TrainingOnly):
    pass


class DistributedBucketSampler(_TrainingOnly):
    pass

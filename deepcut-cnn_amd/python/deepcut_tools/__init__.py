"""Host-side tools of the MI355X DeeperCut path: model-definition generator, .caffemodel writer and
the conditioned synthetic-weight generator used by tests and bench.py (the trained weights are not
shipped with the reference: models/deepercut/download_models.sh fetches them)."""
from .model_zoo import deepercut_prototxt, deepercut_layer_table  # noqa: F401
from .caffemodel import write_caffemodel, read_caffemodel  # noqa: F401
from .synth import synth_weights, write_synth_caffemodel  # noqa: F401
from .shard import lpt_shards, gather_maps, gather_maps_known  # noqa: F401
from .pipeline import Pipeline  # noqa: F401
from .runner import ShardedPoseRunner, plan_work, net_input_shape, rank_batches, group_units  # noqa: F401
from .tuning import tune_in_flight  # noqa: F401

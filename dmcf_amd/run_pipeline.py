"""The reference's ``run_pipeline.py`` for the test split on the MI355X path: YAML config -> model -> checkpoint ->
``get_rollout`` -> ``Simulator.run_rollout`` -> ``write_results`` (run_pipeline.py:80-154, pipelines/simulator.py:111-165).

    python -m dmcf_amd.run_pipeline -c configs/Liquid3d.yml --split test --dataset_path <dir with *.msgpack.zst> \\
        --ckpt_path checkpoints/Liquid3d/ckpt --output_dir output [--model.timestep 0.02 ...]

Same flags as the reference; ``--section.key value`` overrides go through Config.merge_cfg_file.  ``--split train`` /
``valid`` raise: training and the validation metrics are outside the per-step hot path.
"""
import argparse
import random
import sys

import numpy as np


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="Run a network over the test split (run_pipeline.py)")
    parser.add_argument("-c", "--cfg_file", help="path to the config file", required=True)
    parser.add_argument("--dataset_path", help="path to the dataset")
    parser.add_argument("--ckpt_path", help="path to the checkpoint")
    parser.add_argument("--device", help="device to run the pipeline", default="gpu")
    parser.add_argument("--split", help="train or test", default="train")
    parser.add_argument("--regen", default=False, action="store_true")
    parser.add_argument("--restart", default=False, action="store_true")
    parser.add_argument("--main_log_dir", help="the dir to save logs and models")
    parser.add_argument("--output_dir", help="the dir to save outputs")
    args, unknown = parser.parse_known_args(argv)
    extra = argparse.ArgumentParser(description="Extra arguments")  # run_pipeline.py:46-52
    for arg in unknown:
        if arg.startswith(("-", "--")):
            extra.add_argument(arg)
    return args, {k: v for k, v in vars(extra.parse_args(unknown)).items()}


def build(args, extra, data=None):
    """-> the Simulator pipeline of run_pipeline.py:104-121 (``data``: scenes already in memory instead of a dataset_path)."""
    from . import models, pipelines
    from .datasets import DatasetGroup
    from .utils.config import Config
    cfg = Config.load_from_file(args.cfg_file)
    if args.device in ("gpu", None):
        args.device = "cuda"
    Pipeline = getattr(pipelines, cfg.pipeline.name)
    Model = getattr(models, cfg.model.name)
    cfg_dataset, cfg_pipeline, cfg_model = Config.merge_cfg_file(cfg, args, extra)
    dataset = DatasetGroup(**cfg_dataset, split=args.split, regen=args.regen, data=data)
    model = Model(**cfg_model)
    return Pipeline(model, dataset, **cfg_pipeline)


def main(argv=None, data=None):
    random.seed(42)
    np.random.seed(42)
    args, extra = parse_args(argv)
    if args.split != "test":
        raise NotImplementedError(f"--split {args.split}: only the test split (rollout + write_results) is on the hot path")
    pipeline = build(args, extra, data)
    return pipeline.run_test()


if __name__ == "__main__":
    print("\n".join(main(sys.argv[1:])))

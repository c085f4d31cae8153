"""Per-dataset run configs (mirror of reference ofa/imagenet_codebase/run_manager/__init__.py:127-232).
`Div2K_SetXXRunConfig(**args.__dict__)` keeps working; its lazy `data_provider` resolves to the real
DIV2K provider when the dataset directory exists.  A missing directory is an ERROR unless synthetic data was asked
for explicitly (`allow_synthetic=True` or OFASR_ALLOW_SYNTHETIC_DATA=1): a mistyped path must not train on noise."""
import os

from .sr_run_manager import RunConfig, SRRunManager  # noqa: F401
from ..data_providers.synthetic_sr import SyntheticSRDataProvider


class SyntheticSRRunConfig(RunConfig):

    def __init__(self, n_epochs=1, init_lr=1e-3, lr_schedule_type="cosine", lr_schedule_param=None,
                 dataset="synthetic_sr", train_batch_size=16, test_batch_size=1, valid_size=None, opt_type="adam",
                 opt_param=None, weight_decay=3e-5, label_smoothing=0.0, no_decay_keys="bn#bias", mixup_alpha=None,
                 model_init="he_fout", validation_frequency=1, print_frequency=10, n_worker=0, image_size=256,
                 n_train_batches=4, n_test_batches=2, data_seed=0, test_sizes=None, **kwargs):
        super().__init__(n_epochs, init_lr, lr_schedule_type, lr_schedule_param, dataset, train_batch_size,
                         test_batch_size, valid_size, opt_type, opt_param, weight_decay, label_smoothing,
                         no_decay_keys, mixup_alpha, model_init, validation_frequency, print_frequency)
        self.n_worker = n_worker
        self.image_size = image_size
        self._synthetic = dict(n_train_batches=n_train_batches, n_test_batches=n_test_batches, seed=data_seed,
                               test_sizes=test_sizes)

    @property
    def data_provider(self):
        if self.__dict__.get("_data_provider", None) is None:
            from ... import distributed as dd
            self.__dict__["_data_provider"] = SyntheticSRDataProvider(
                train_batch_size=self.train_batch_size, test_batch_size=self.test_batch_size,
                image_size=self.image_size, rank=dd.rank(), num_replicas=dd.world_size(), **self._synthetic)
        return self.__dict__["_data_provider"]


class Div2K_SetXXRunConfig(SyntheticSRRunConfig):
    """reference :127-160.  `data_provider` is the real Div2K_SetXXDataProvider (data_providers/div2k_setxx.py)
    when `$OFASR_DIV2K_ROOT` (default /SSD/div2k_setxx, the reference's DEFAULT_PATH) holds `train/` and `val/`.
    Otherwise it raises, unless the caller opted into the synthetic provider of the same interface
    (`allow_synthetic=True` / OFASR_ALLOW_SYNTHETIC_DATA=1: the DIV2K/SetXX files are absent in this environment);
    the provider actually used is recorded in `self.dataset`, hence in run.config and in every checkpoint."""

    def __init__(self, n_epochs=150, init_lr=0.05, lr_schedule_type="cosine", lr_schedule_param=None,
                 dataset="div2k_setxx", train_batch_size=256, test_batch_size=500, valid_size=None, opt_type="sgd",
                 opt_param=None, weight_decay=4e-5, label_smoothing=0.1, no_decay_keys=None, mixup_alpha=None,
                 model_init="he_fout", validation_frequency=1, print_frequency=10, n_worker=32,
                 resize_scale=0.08, distort_color=None, image_size=32, allow_synthetic=None, **kwargs):
        super().__init__(n_epochs, init_lr, lr_schedule_type, lr_schedule_param, dataset, train_batch_size,
                         test_batch_size, valid_size, opt_type, opt_param, weight_decay, label_smoothing,
                         no_decay_keys, mixup_alpha, model_init, validation_frequency, print_frequency,
                         n_worker=n_worker, image_size=image_size,
                         **{k: v for k, v in kwargs.items() if k in ("n_train_batches", "n_test_batches",
                                                                     "data_seed", "test_sizes")})
        self.resize_scale = resize_scale
        self.distort_color = distort_color
        self.dataset_root = os.environ.get("OFASR_DIV2K_ROOT", "/SSD/div2k_setxx")
        self.allow_synthetic = (os.environ.get("OFASR_ALLOW_SYNTHETIC_DATA", "0") == "1") if allow_synthetic is None \
            else bool(allow_synthetic)

    @property
    def data_provider(self):
        if self.__dict__.get("_data_provider", None) is None:
            root = self.dataset_root
            if os.path.isdir(os.path.join(root, "train")) and os.path.isdir(os.path.join(root, "val")):
                from ... import distributed as dd
                from ..data_providers.div2k_setxx import Div2K_SetXXDataProvider
                ws = dd.world_size()
                self.__dict__["_data_provider"] = Div2K_SetXXDataProvider(
                    save_path=root, train_batch_size=self.train_batch_size, test_batch_size=self.test_batch_size,
                    valid_size=self.valid_size, n_worker=self.n_worker, resize_scale=self.resize_scale,
                    distort_color=self.distort_color, image_size=self.image_size,
                    num_replicas=ws if ws > 1 else None, rank=dd.rank() if ws > 1 else None)
            elif self.allow_synthetic:
                from ... import distributed as dd
                if dd.rank() == 0:
                    print("WARNING: %s has no train/ + val/ -- training on SYNTHETIC images (allow_synthetic)" % root,
                          flush=True)
                self.dataset = "synthetic_sr (stand-in for div2k_setxx: %s missing)" % root
                return super().data_provider
            else:
                raise FileNotFoundError(
                    "DIV2K/SetXX dataset not found: %s must hold train/ and val/ (set OFASR_DIV2K_ROOT).  Pass "
                    "allow_synthetic=True / --synthetic or set OFASR_ALLOW_SYNTHETIC_DATA=1 to run on synthetic "
                    "images instead." % root)
        return self.__dict__["_data_provider"]

from .test_dataset_gram import TestDatasetGRAM  # noqa: F401

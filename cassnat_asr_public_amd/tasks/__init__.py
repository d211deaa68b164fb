"""Mirror of the reference's ``tasks`` package for the accelerated path (src/tasks/__init__.py)."""
from .base_task import BaseTask  # noqa: F401
from .cassnat_task import CassNATTask  # noqa: F401
from .art_task import ArtTask  # noqa: F401

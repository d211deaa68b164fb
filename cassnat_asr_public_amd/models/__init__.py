"""Mirror of the reference's ``models`` package surface for the accelerated path (src/models/__init__.py:5)."""
from .cassnat import make_model as make_cassnat_model  # noqa: F401
from .transformer import make_model as make_transformer  # noqa: F401  (src/models/__init__.py:2)

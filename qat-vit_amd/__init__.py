"""qat-vit_amd: MI355X-native QAT-ViT student training path (drop-in below the
reference's ``src.models`` API).  Import as ``qat_vit_amd``."""
from .model_registry import (  # noqa: F401
    PLATFORM,
    QATWrapper,
    create_model,
    create_student,
    create_teacher,
    list_available_models,
    register_model,
)

from .export import Int8Student, export_int8, import_int8  # noqa: F401,E402
from .optim import ClipAdamW  # noqa: F401,E402

__all__ = ["ClipAdamW", "Int8Student", "export_int8", "import_int8", "PLATFORM", "QATWrapper", "create_model", "create_student", "create_teacher", "list_available_models", "register_model"]

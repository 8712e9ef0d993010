from .pose_metrics import BasicKeyPointDecoder, GaussTaylorKeyPointDecoder  # noqa: F401

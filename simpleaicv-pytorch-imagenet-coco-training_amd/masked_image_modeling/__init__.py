"""Drop-in mirrors of simpleAICV.masked_image_modeling.{models, losses} for the ViT MAE pre-training recipe
(BASELINE config #5), backed by the HIP engine (mae_engine.py)."""

"""Drop-in mirrors of simpleAICV.detection.{models, losses, decode} backed by the HIP engine."""

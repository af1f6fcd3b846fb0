"""Drop-in mirrors of simpleAICV.classification.{backbones, losses} for the ResNet50 ImageNet recipe
(SURVEY.md 8f-1), backed by the HIP engine."""

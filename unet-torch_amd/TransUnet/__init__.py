"""MI355X-native TransUNet (R50-ViT hybrid): drop-in for the reference `TransUnet` package."""

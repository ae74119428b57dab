"""Recipe registry keyed by ``config.net.kind`` (reference scripts/resources.py:55-83)."""
from .types import RECIPE_VERSION, ModelRecipe


def get_recipe(kind: str) -> ModelRecipe:
    from . import duo_vanilla_bert, duo_vanilla_vit, froyo_bert, froyo_vit, ltt_bert, ltt_vit, vanilla_bert, vanilla_vit
    table = {
        "vanilla_vit": vanilla_vit.vanilla_vit_recipe, "vanilla_bert": vanilla_bert.vanilla_bert_recipe,
        "duo_vanilla_vit": duo_vanilla_vit.duo_vanilla_vit_recipe, "duo_vanilla_bert": duo_vanilla_bert.duo_vanilla_bert_recipe,
        "froyo_vit": froyo_vit.froyo_vit_recipe, "froyo_bert": froyo_bert.froyo_bert_recipe,
        "ltt_vit": ltt_vit.ltt_vit_recipe, "ltt_bert": ltt_bert.ltt_bert_recipe,
    }
    if kind not in table:
        raise ValueError(f"unsupported net.kind: {kind} (the kernel_shap recipes are outside this build's scope)")
    recipe = table[kind]()
    if recipe.version != RECIPE_VERSION:
        raise ValueError(f"recipe version mismatch: {recipe.version} != {RECIPE_VERSION}")
    return recipe

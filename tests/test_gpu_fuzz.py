"""Random scene graphs (nested lists, BVHs over mixed children, boxes, transform chains, media) against the oracle,
both traversal orders, incoherent batches and batches in which whole waves trace the same ray."""
import pytest

pytestmark = pytest.mark.gpu


def test_random_scene_graphs_match_the_oracle():
    import fuzz_scenes
    for seed in (2, 7):  # seed 2, scene 54 holds a RotateY angle at which glibc's sincos() and sin()/cos() differ in the last place
        done, failed = fuzz_scenes.run(80, seed)
        assert done >= 70 and not failed, failed


def test_random_scene_graphs_with_nested_and_textured_media():
    """The same, with media whose boundary is any node (a medium inside it included) and, in every other scene, phase
    functions over an image texture, whose medium hits must report the u,v the reference's record would hold."""
    import fuzz_scenes
    done, failed = fuzz_scenes.run(60, 11, rich_media=True)
    assert done >= 40 and not failed, failed


def test_random_scenes_render_like_the_oracle():
    """Random scene graphs with random materials (diffuse, checker, metal, glass, media), light lists of a quad and
    sometimes a sphere, random cameras: at most three pixels of a frame may differ from the oracle beyond rounding."""
    import fuzz_render
    done, failed = fuzz_render.run(60, 5)
    assert done >= 50 and not failed, failed
    # seed 72, scene 314: paths that start inside a spherical light (light pdf 0/0) and end on a light's black
    # back face; the oracle's samples are NaN there, and nine pixels once were finite on the device
    done, failed = fuzz_render.run(315, 72, only=314)
    assert done == 1 and not failed, failed

"""Scene scripts: the scene definitions of the reference's main.rs restated against the host
mirror, plus the synthetic closest-hit workloads (random-sphere and triangle-soup scenes,
ray batches).  Each function returns (world, lights, camera); `lights` is None where the
reference scene predates the light list.

Random choices use the seeded scene-construction stream (rtweekend.seed), so every rank
and the CPU oracle see the same scene.  Assets the reference names but does not ship
(earthmap.jpg, grumble.jpg, models/*.obj) are replaced by procedural stand-ins of the
same kind (an RGB8 image, a triangle array).
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np

from . import rtweekend
from .camera import Camera
from .hittable import (BvhNode, ConstantMedium, Hittable, HittableList, Mesh, Quad, RotateY, Sphere, SphereSet, Translate,
                       Triangle, box_new)
from .material import Dielectric, DiffuseLight, Lambertian, Metal
from .rtweekend import random_double, random_double_range
from .texture import CheckerTexture, ImageTexture, NoiseTexture
from .vec3 import Color, Point3, Vec3

SceneTuple = Tuple[Hittable, Optional[Hittable], Camera]


def procedural_image(width: int = 256, height: int = 128) -> np.ndarray:
    """RGB8 stand-in for earthmap.jpg / grumble.jpg (main.rs:245,734,752): smooth bands + a grid."""
    y, x = np.mgrid[0:height, 0:width]
    r = (127.5 * (1 + np.sin(x * (2 * math.pi / width) * 3))).astype(np.uint8)
    g = (127.5 * (1 + np.cos(y * (2 * math.pi / height) * 2))).astype(np.uint8)
    b = (((x // 16 + y // 16) % 2) * 200 + 30).astype(np.uint8)
    return np.stack([r, g, b], axis=-1)


def for_output13() -> SceneTuple:  # main.rs:41-97
    world = HittableList()
    world.add(Sphere(Point3(0.0, -100.5, -1.0), 100.0, Lambertian(Color(0.8, 0.8, 0.0))))
    world.add(Sphere(Point3(0.0, 0.0, -1.2), 0.5, Lambertian(Color(0.1, 0.2, 0.5))))
    world.add(Sphere(Point3(-1.0, 0.0, -1.0), 0.5, Dielectric(1.50)))
    world.add(Sphere(Point3(-1.0, 0.0, -1.0), 0.4, Dielectric(1.00 / 1.50)))
    world.add(Sphere(Point3(1.0, 0.0, -1.0), 0.5, Metal(Color(0.8, 0.6, 0.2), 1.0)))
    cam = Camera()
    cam.aspect_ratio, cam.image_width, cam.samples_per_pixel, cam.max_depth = 16.0 / 9.0, 400, 100, 50
    cam.background = Color(0.70, 0.80, 1.00)  # the scene dates from the sky-gradient revisions
    cam.vfov, cam.lookfrom, cam.lookat, cam.vup = 20.0, Point3(-2.0, 2.0, 1.0), Point3(0.0, 0.0, -1.0), Vec3(0.0, 1.0, 0.0)
    cam.defocus_angle, cam.focus_dist = 10.0, 3.4
    return world, None, cam


def book1_final(seed: int = 0, moving: bool = False, checker_ground: bool = False, image_width: int = 1200,
                samples_per_pixel: int = 10, max_depth: int = 50) -> SceneTuple:
    """last_picture_the_first_book (main.rs:99-202).  Defaults give the Book-1 picture (static
    spheres, grey ground: the commented variant main.rs:102-107,143); moving/checker_ground
    select the live Book-2 edits (main.rs:109-142)."""
    rtweekend.seed(seed)
    world = HittableList()
    if checker_ground:
        ground = Lambertian.from_texture(CheckerTexture.from_colors(0.32, Color(0.2, 0.3, 0.1), Color(0.9, 0.9, 0.9)))
    else:
        ground = Lambertian(Color(0.5, 0.5, 0.5))
    world.add(Sphere(Point3(0.0, -1000.0, 0.0), 1000.0, ground))
    for a in range(-11, 11):
        for b in range(-11, 11):
            choose_mat = random_double()
            cx = a + 0.9 * random_double()
            cz = b + 0.9 * random_double()
            center = Point3(cx, 0.2, cz)
            if (center - Point3(4.0, 0.2, 0.0)).length() > 0.9:
                if choose_mat < 0.8:
                    albedo = Color.random() * Color.random()
                    mat = Lambertian(albedo)
                    if moving:
                        center2 = center + Vec3(0.0, random_double_range(0.0, 0.5), 0.0)
                        world.add(Sphere.new_moving(center, center2, 0.2, mat))
                    else:
                        world.add(Sphere(center, 0.2, mat))
                elif choose_mat < 0.95:
                    albedo = Color.random_range(0.5, 1.0)
                    fuzz = random_double_range(0.0, 0.5)
                    world.add(Sphere(center, 0.2, Metal(albedo, fuzz)))
                else:
                    world.add(Sphere(center, 0.2, Dielectric(1.5)))
    world.add(Sphere(Point3(0.0, 1.0, 0.0), 1.0, Dielectric(1.5)))
    world.add(Sphere(Point3(-4.0, 1.0, 0.0), 1.0, Lambertian(Color(0.4, 0.2, 0.1))))
    world.add(Sphere(Point3(4.0, 1.0, 0.0), 1.0, Metal(Color(0.7, 0.6, 0.5), 0.0)))
    world_bvh = HittableList()
    world_bvh.add(BvhNode(world))  # main.rs:178-180
    cam = Camera()
    cam.aspect_ratio, cam.image_width, cam.samples_per_pixel, cam.max_depth = 16.0 / 9.0, image_width, samples_per_pixel, max_depth
    cam.background = Color(0.70, 0.80, 1.00)
    cam.vfov, cam.lookfrom, cam.lookat, cam.vup = 20.0, Point3(13.0, 2.0, 3.0), Point3(0.0, 0.0, 0.0), Vec3(0.0, 1.0, 0.0)
    cam.defocus_angle, cam.focus_dist = 0.6, 10.0
    return world_bvh, None, cam


def _sky_camera(vfov: float, lookfrom: Point3, lookat: Point3, aspect: float = 16.0 / 9.0) -> Camera:
    cam = Camera()
    cam.aspect_ratio, cam.image_width, cam.samples_per_pixel, cam.max_depth = aspect, 400, 100, 50
    cam.background = Color(0.70, 0.80, 1.00)
    cam.vfov, cam.lookfrom, cam.lookat, cam.vup = vfov, lookfrom, lookat, Vec3(0.0, 1.0, 0.0)
    cam.defocus_angle = 0.0
    return cam


def checkered_spheres() -> SceneTuple:  # main.rs:204-242
    world = HittableList()
    mat = Lambertian.from_texture(CheckerTexture.from_colors(0.32, Color(0.2, 0.3, 0.1), Color(0.9, 0.9, 0.9)))
    world.add(Sphere(Point3(0.0, -10.0, 0.0), 10.0, mat))
    world.add(Sphere(Point3(0.0, 10.0, 0.0), 10.0, mat))
    return world, None, _sky_camera(20.0, Point3(13.0, 2.0, 3.0), Point3(0.0, 0.0, 0.0))


def earth(image_file: str | None = None) -> SceneTuple:  # main.rs:244-268
    """`image_file` is looked up the way main.rs:245 looks up "earthmap.jpg" (RtwImage.load); the reference does
    not ship that file, so without a name the globe carries a procedural image of the same role."""
    world = HittableList()
    tex = ImageTexture(image_file) if image_file is not None else ImageTexture(procedural_image())
    world.add(Sphere(Point3(0.0, 0.0, 0.0), 2.0, Lambertian.from_texture(tex)))
    return world, None, _sky_camera(20.0, Point3(0.0, 0.0, 12.0), Point3(0.0, 0.0, 0.0))


def perlin_spheres(seed: int = 0) -> SceneTuple:  # main.rs:270-303
    rtweekend.seed(seed)
    world = HittableList()
    mat = Lambertian.from_texture(NoiseTexture(4.0))
    world.add(Sphere(Point3(0.0, -1000.0, 0.0), 1000.0, mat))
    world.add(Sphere(Point3(0.0, 2.0, 0.0), 2.0, mat))
    return world, None, _sky_camera(20.0, Point3(13.0, 2.0, 3.0), Point3(0.0, 0.0, 0.0))


def quads() -> SceneTuple:  # main.rs:305-366
    world = HittableList()
    world.add(Quad(Point3(-3.0, -2.0, 5.0), Vec3(0.0, 0.0, -4.0), Vec3(0.0, 4.0, 0.0), Lambertian(Color(1.0, 0.2, 0.2))))
    world.add(Quad(Point3(-2.0, -2.0, 0.0), Vec3(4.0, 0.0, 0.0), Vec3(0.0, 4.0, 0.0), Lambertian(Color(0.2, 1.0, 0.2))))
    world.add(Quad(Point3(3.0, -2.0, 1.0), Vec3(0.0, 0.0, 4.0), Vec3(0.0, 4.0, 0.0), Lambertian(Color(0.2, 0.2, 1.0))))
    world.add(Quad(Point3(-2.0, 3.0, 1.0), Vec3(4.0, 0.0, 0.0), Vec3(0.0, 0.0, 4.0), Lambertian(Color(1.0, 0.5, 0.0))))
    world.add(Quad(Point3(-2.0, -3.0, 5.0), Vec3(4.0, 0.0, 0.0), Vec3(0.0, 0.0, -4.0), Lambertian(Color(0.2, 0.8, 0.8))))
    world_bvh = HittableList()
    world_bvh.add(BvhNode(world))
    return world_bvh, None, _sky_camera(80.0, Point3(0.0, 0.0, 9.0), Point3(0.0, 0.0, 0.0), aspect=1.0)


def simple_light(seed: int = 0, sphere_light: bool = True) -> SceneTuple:  # main.rs:368-419
    """`sphere_light` = False: the scene one step earlier in the reference's history, lit by the rectangle alone
    (time_record.txt:15, output29.ppm); with the sphere above it is the scene as it stands (output30.ppm)."""
    rtweekend.seed(seed)
    world = HittableList()
    mat = Lambertian.from_texture(NoiseTexture(4.0))
    world.add(Sphere(Point3(0.0, -1000.0, 0.0), 1000.0, mat))
    world.add(Sphere(Point3(0.0, 2.0, 0.0), 2.0, mat))
    difflight = DiffuseLight.from_color(Color(4.0, 4.0, 4.0))
    if sphere_light:
        world.add(Sphere(Point3(0.0, 7.0, 0.0), 2.0, difflight))
    world.add(Quad(Point3(3.0, 1.0, -2.0), Vec3(2.0, 0.0, 0.0), Vec3(0.0, 2.0, 0.0), difflight))
    cam = _sky_camera(20.0, Point3(26.0, 3.0, 6.0), Point3(0.0, 2.0, 0.0))
    cam.background = Color(0.0, 0.0, 0.0)
    return world, None, cam


def _cornell_camera(spp: int) -> Camera:
    cam = Camera()
    cam.aspect_ratio, cam.image_width, cam.samples_per_pixel, cam.max_depth = 1.0, 600, spp, 50
    cam.background = Color(0.0, 0.0, 0.0)
    cam.vfov, cam.lookfrom, cam.lookat, cam.vup = 40.0, Point3(278.0, 278.0, -800.0), Point3(278.0, 278.0, 0.0), Vec3(0.0, 1.0, 0.0)
    cam.defocus_angle = 0.0
    return cam


def cornell_box(image_width: int = 600, samples_per_pixel: int = 1000, max_depth: int = 50) -> SceneTuple:
    """cornell_box (main.rs:421-551): the only scene live at HEAD.  Flat list, glass sphere,
    lights = {ceiling quad, glass sphere}."""
    world = HittableList()
    red = Lambertian(Color(0.65, 0.05, 0.05))
    white = Lambertian(Color(0.73, 0.73, 0.73))
    green = Lambertian(Color(0.12, 0.45, 0.15))
    light = DiffuseLight.from_color(Color(15.0, 15.0, 15.0))
    world.add(Quad(Point3(555.0, 0.0, 0.0), Vec3(0.0, 555.0, 0.0), Vec3(0.0, 0.0, 555.0), green))
    world.add(Quad(Point3(0.0, 0.0, 0.0), Vec3(0.0, 555.0, 0.0), Vec3(0.0, 0.0, 555.0), red))
    world.add(Quad(Point3(0.0, 0.0, 0.0), Vec3(555.0, 0.0, 0.0), Vec3(0.0, 0.0, 555.0), white))
    world.add(Quad(Point3(555.0, 555.0, 555.0), Vec3(-555.0, 0.0, 0.0), Vec3(0.0, 0.0, -555.0), white))
    world.add(Quad(Point3(0.0, 0.0, 555.0), Vec3(555.0, 0.0, 0.0), Vec3(0.0, 555.0, 0.0), white))
    world.add(Quad(Point3(343.0, 554.0, 332.0), Vec3(-130.0, 0.0, 0.0), Vec3(0.0, 0.0, -105.0), light))
    box1: Hittable = box_new(Point3(0.0, 0.0, 0.0), Point3(165.0, 330.0, 165.0), white)
    box1 = RotateY(box1, 15.0)
    box1 = Translate(box1, Vec3(265.0, 0.0, 295.0))
    world.add(box1)
    world.add(Sphere(Point3(190.0, 90.0, 190.0), 90.0, Dielectric(1.5)))
    empty_material = Lambertian(Color(0.73, 0.73, 0.73))
    lights = HittableList()
    lights.add(Quad(Point3(343.0, 554.0, 332.0), Vec3(-130.0, 0.0, 0.0), Vec3(0.0, 0.0, -105.0), empty_material))
    lights.add(Sphere(Point3(190.0, 90.0, 190.0), 90.0, empty_material))
    cam = _cornell_camera(samples_per_pixel)
    cam.image_width, cam.max_depth = image_width, max_depth
    return world, lights, cam


def cornell_two_blocks(image_width: int = 600, samples_per_pixel: int = 200, max_depth: int = 50, blocks: str = "rotated") -> SceneTuple:
    """The Cornell box of Book 2 as the reference's history rendered it, before the light list existed (material PDF
    alone: lights = None).  blocks = "rotated": the "standard Cornell box scene" (output33.ppm, time_record.txt:19), the
    room of cornell_box with two rotated blocks and no glass sphere; "axis": the step before, the same two blocks not
    yet rotated ("Cornell box with two blocks", output32.ppm, time_record.txt:18); "none": the empty room
    (output31.ppm, time_record.txt:17)."""
    world = HittableList()
    red = Lambertian(Color(0.65, 0.05, 0.05))
    white = Lambertian(Color(0.73, 0.73, 0.73))
    green = Lambertian(Color(0.12, 0.45, 0.15))
    light = DiffuseLight.from_color(Color(15.0, 15.0, 15.0))
    world.add(Quad(Point3(555.0, 0.0, 0.0), Vec3(0.0, 555.0, 0.0), Vec3(0.0, 0.0, 555.0), green))
    world.add(Quad(Point3(0.0, 0.0, 0.0), Vec3(0.0, 555.0, 0.0), Vec3(0.0, 0.0, 555.0), red))
    world.add(Quad(Point3(343.0, 554.0, 332.0), Vec3(-130.0, 0.0, 0.0), Vec3(0.0, 0.0, -105.0), light))
    world.add(Quad(Point3(0.0, 0.0, 0.0), Vec3(555.0, 0.0, 0.0), Vec3(0.0, 0.0, 555.0), white))
    world.add(Quad(Point3(555.0, 555.0, 555.0), Vec3(-555.0, 0.0, 0.0), Vec3(0.0, 0.0, -555.0), white))
    world.add(Quad(Point3(0.0, 0.0, 555.0), Vec3(555.0, 0.0, 0.0), Vec3(0.0, 555.0, 0.0), white))
    if blocks == "rotated":
        box1: Hittable = box_new(Point3(0.0, 0.0, 0.0), Point3(165.0, 330.0, 165.0), white)
        world.add(Translate(RotateY(box1, 15.0), Vec3(265.0, 0.0, 295.0)))
        box2: Hittable = box_new(Point3(0.0, 0.0, 0.0), Point3(165.0, 165.0, 165.0), white)
        world.add(Translate(RotateY(box2, -18.0), Vec3(130.0, 0.0, 65.0)))
    elif blocks == "axis":
        world.add(box_new(Point3(130.0, 0.0, 65.0), Point3(295.0, 165.0, 230.0), white))
        world.add(box_new(Point3(265.0, 0.0, 295.0), Point3(430.0, 330.0, 460.0), white))
    elif blocks != "none":
        raise ValueError(blocks)
    cam = _cornell_camera(samples_per_pixel)
    cam.image_width, cam.max_depth = image_width, max_depth
    return world, None, cam


def cornell_smoke(image_width: int = 600, samples_per_pixel: int = 200, max_depth: int = 50) -> SceneTuple:
    """cornell_smoke (main.rs:553-649): two constant media bounded by instanced boxes, inside a BVH."""
    world = HittableList()
    red = Lambertian(Color(0.65, 0.05, 0.05))
    white = Lambertian(Color(0.73, 0.73, 0.73))
    green = Lambertian(Color(0.12, 0.45, 0.15))
    light = DiffuseLight.from_color(Color(7.0, 7.0, 7.0))
    world.add(Quad(Point3(555.0, 0.0, 0.0), Vec3(0.0, 555.0, 0.0), Vec3(0.0, 0.0, 555.0), green))
    world.add(Quad(Point3(0.0, 0.0, 0.0), Vec3(0.0, 555.0, 0.0), Vec3(0.0, 0.0, 555.0), red))
    world.add(Quad(Point3(113.0, 554.0, 127.0), Vec3(333.0, 0.0, 0.0), Vec3(0.0, 0.0, 305.0), light))
    world.add(Quad(Point3(0.0, 555.0, 0.0), Vec3(555.0, 0.0, 0.0), Vec3(0.0, 0.0, 555.0), white))
    world.add(Quad(Point3(0.0, 0.0, 0.0), Vec3(555.0, 0.0, 0.0), Vec3(0.0, 0.0, 555.0), white))
    world.add(Quad(Point3(0.0, 0.0, 555.0), Vec3(555.0, 0.0, 0.0), Vec3(0.0, 555.0, 0.0), white))
    box1: Hittable = box_new(Point3(0.0, 0.0, 0.0), Point3(165.0, 330.0, 165.0), white)
    box1 = Translate(RotateY(box1, 15.0), Vec3(265.0, 0.0, 295.0))
    box2: Hittable = box_new(Point3(0.0, 0.0, 0.0), Point3(165.0, 165.0, 165.0), white)
    box2 = Translate(RotateY(box2, -18.0), Vec3(130.0, 0.0, 65.0))
    world.add(ConstantMedium.new_with_color(box1, 0.01, Color(0.0, 0.0, 0.0)))
    world.add(ConstantMedium.new_with_color(box2, 0.01, Color(1.0, 1.0, 1.0)))
    world_bvh = HittableList()
    world_bvh.add(BvhNode(world))
    lights = HittableList()  # the scene predates the light list; the light quad is what later revisions pass
    lights.add(Quad(Point3(113.0, 554.0, 127.0), Vec3(333.0, 0.0, 0.0), Vec3(0.0, 0.0, 305.0), white))
    cam = _cornell_camera(samples_per_pixel)
    cam.image_width, cam.max_depth = image_width, max_depth
    return world_bvh, lights, cam


def book2_final(image_width: int = 800, samples_per_pixel: int = 100, max_depth: int = 40, seed: int = 0,
                with_lights: bool = False) -> SceneTuple:
    """final_scene (main.rs:651-788).  The reference calls it with (800, 10000, 40) and (400, 250, 4)."""
    rtweekend.seed(seed)
    boxes1 = HittableList()
    ground = Lambertian(Color(0.48, 0.83, 0.53))
    boxes_per_side = 20
    for i in range(boxes_per_side):
        for j in range(boxes_per_side):
            w = 100.0
            x0 = -1000.0 + i * w
            z0 = -1000.0 + j * w
            y0 = 0.0
            x1 = x0 + w
            y1 = random_double_range(1.0, 101.0)
            z1 = z0 + w
            boxes1.add(box_new(Point3(x0, y0, z0), Point3(x1, y1, z1), ground))
    world = HittableList()
    world.add(BvhNode(boxes1))
    light = DiffuseLight.from_color(Color(7.0, 7.0, 7.0))
    light_quad = (Point3(123.0, 554.0, 147.0), Vec3(300.0, 0.0, 0.0), Vec3(0.0, 0.0, 265.0))
    world.add(Quad(*light_quad, light))
    center1 = Point3(400.0, 400.0, 200.0)
    center2 = center1 + Vec3(30.0, 0.0, 0.0)
    world.add(Sphere.new_moving(center1, center2, 50.0, Lambertian(Color(0.7, 0.3, 0.1))))
    world.add(Sphere(Point3(260.0, 150.0, 45.0), 50.0, Dielectric(1.5)))
    world.add(Sphere(Point3(0.0, 150.0, 145.0), 50.0, Metal(Color(0.8, 0.8, 0.9), 1.0)))
    boundary = Sphere(Point3(360.0, 150.0, 145.0), 70.0, Dielectric(1.5))
    world.add(boundary)
    world.add(ConstantMedium.new_with_color(boundary, 0.2, Color(0.2, 0.4, 0.9)))
    boundary2 = Sphere(Point3(0.0, 0.0, 0.0), 5000.0, Dielectric(1.5))
    world.add(ConstantMedium.new_with_color(boundary2, 0.0001, Color(1.0, 1.0, 1.0)))
    image = procedural_image()
    world.add(Sphere(Point3(400.0, 200.0, 400.0), 100.0, Lambertian.from_texture(ImageTexture(image))))
    world.add(Sphere(Point3(220.0, 280.0, 300.0), 80.0, Lambertian.from_texture(NoiseTexture(0.2))))
    boxes2 = HittableList()
    grumble = Lambertian.from_texture(ImageTexture(image))
    for _ in range(1000):
        boxes2.add(Sphere(Point3.random_range(0.0, 165.0), 10.0, grumble))
    world.add(Translate(RotateY(BvhNode(boxes2), 15.0), Vec3(-100.0, 270.0, 395.0)))
    lights = None
    if with_lights:
        lights = HittableList()
        lights.add(Quad(*light_quad, ground))
    cam = Camera()
    cam.aspect_ratio, cam.image_width, cam.samples_per_pixel, cam.max_depth = 1.0, image_width, samples_per_pixel, max_depth
    cam.background = Color(0.0, 0.0, 0.0)
    cam.vfov, cam.lookfrom, cam.lookat, cam.vup = 40.0, Point3(478.0, 278.0, -600.0), Point3(278.0, 278.0, 0.0), Vec3(0.0, 1.0, 0.0)
    cam.defocus_angle = 0.0
    return world, lights, cam


def test_triangle() -> SceneTuple:  # main.rs:864-895
    world = HittableList()
    world.add(Triangle(Point3(-10.0, 0.0, -5.0), Point3(10.0, 0.0, -5.0), Point3(0.0, 10.0, -5.0), Lambertian(Color(0.8, 0.2, 0.2))))
    world.add(Sphere(Point3(0.0, 0.0, -20.0), 10.0, Lambertian(Color(0.2, 0.2, 1.0))))
    return world, None, _sky_camera(80.0, Point3(0.0, 0.0, 9.0), Point3(0.0, 0.0, 0.0), aspect=1.0)


def mesh_and_spheres(seed: int = 0, subdiv: int = 6, image_width: int = 4096, samples_per_pixel: int = 4096,
                     max_depth: int = 50) -> SceneTuple:
    """The multi-GPU frame of the measurement plan: the Book-1 sphere field (BVH, main.rs:99-180) around
    a bunny-class closed mesh (Mesh = BVH over triangles, mesh.rs:163) standing on the ground, lit by
    the sky background.  4096 x 2304 at 4096 spp by default; tests render it small."""
    world_bvh, _, cam = book1_final(seed=seed)
    tris = icosphere_mesh(subdiv, seed=4) * 3.0 + np.array([-1.5, 0.1, -3.5])  # radius ~1.2, centre (0, 1.6, -2)
    world = HittableList()
    world.add(world_bvh.objects[0])
    world.add(Mesh(tris, Metal(Color(0.8, 0.8, 0.9), 0.05)))
    cam.image_width, cam.samples_per_pixel, cam.max_depth = image_width, samples_per_pixel, max_depth
    cam.lookat = Point3(0.0, 0.8, -1.0)
    return world, None, cam


# ------------------------------------------------------------------ synthetic workloads
def _philox_uniform(n: int, seed: int, dims: int) -> np.ndarray:
    """(n, dims) uniforms in [0,1) from numpy's Philox bit generator (fast bulk stream for
    synthetic inputs; scene scripts above use the library's own stream)."""
    return np.random.Generator(np.random.Philox(key=seed)).random((n, dims))


def triangle_soup(n_triangles: int = 1 << 20, seed: int = 2, edge: float = 0.01) -> np.ndarray:
    """(N, 3, 3) vertices: centroid U[0,1]^3, two edge vectors U[-1,1]^3 * edge."""
    u = _philox_uniform(n_triangles, seed, 9)
    c = u[:, 0:3]
    e1 = (2.0 * u[:, 3:6] - 1.0) * edge
    e2 = (2.0 * u[:, 6:9] - 1.0) * edge
    v0 = c - (e1 + e2) / 3.0
    return np.ascontiguousarray(np.stack([v0, v0 + e1, v0 + e2], axis=1))


def triangle_soup_scene(n_triangles: int = 1 << 20, seed: int = 2) -> Hittable:
    """The 1M-triangle mesh of the closest-hit benchmark: Mesh = BVH over triangles (mesh.rs:163)."""
    return Mesh(triangle_soup(n_triangles, seed), Lambertian(Color(0.5, 0.5, 0.5)))


def icosphere_mesh(subdiv: int = 6, seed: int = 4, displacement: float = 0.08) -> np.ndarray:
    """Closed 'bunny-class' mesh: icosphere (20 * 4^subdiv triangles; 81,920 at subdiv 6) with a
    smooth radial displacement, standing in for the OBJ assets the reference does not ship."""
    t = (1.0 + math.sqrt(5.0)) / 2.0
    verts = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
             (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    faces = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
             (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10),
             (8, 6, 7), (9, 8, 1)]
    v = np.array(verts, dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array(faces, dtype=np.int64)
    for _ in range(subdiv):
        a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
        ab, bc, ca = a + b, b + c, c + a
        ab /= np.linalg.norm(ab, axis=1, keepdims=True)
        bc /= np.linalg.norm(bc, axis=1, keepdims=True)
        ca /= np.linalg.norm(ca, axis=1, keepdims=True)
        tris = np.concatenate([np.stack([a, ab, ca], 1), np.stack([b, bc, ab], 1), np.stack([c, ca, bc], 1),
                               np.stack([ab, bc, ca], 1)])
        v = tris.reshape(-1, 3)
        f = np.arange(len(v)).reshape(-1, 3)
    g = np.random.Generator(np.random.Philox(key=seed))
    k = g.normal(size=(6, 3)) * 3.0
    ph = g.uniform(0, 2 * math.pi, 6)
    r = 1.0 + displacement * np.sum(np.sin(v @ k.T + ph), axis=1) / 6.0
    return np.ascontiguousarray((v * r[:, None]).reshape(-1, 3, 3) * 0.4 + 0.5)


def random_spheres(n: int = 10000, seed: int = 1, extent: float = 1.0) -> SphereSet:
    """S-spheres: centres U[-1,1]^3 * extent, radii U[0.2,1] * extent / cbrt(n) * 0.5."""
    u = _philox_uniform(n, seed, 4)
    c = (2.0 * u[:, 0:3] - 1.0) * extent
    r = (0.2 + 0.8 * u[:, 3]) * extent / n ** (1.0 / 3.0) * 0.5
    return SphereSet(c, r, Lambertian(Color(0.5, 0.5, 0.5)), bvh=True)


def incoherent_rays(n: int, bounds, seed: int = 3):
    """Batch (B): origins uniform in the scene box, directions uniform on the sphere, time U[0,1)."""
    u = _philox_uniform(n, seed, 6)
    lo = np.array([bounds[0], bounds[2], bounds[4]])
    hi = np.array([bounds[1], bounds[3], bounds[5]])
    o = lo + u[:, 0:3] * (hi - lo)
    z = 2.0 * u[:, 3] - 1.0
    phi = 2.0 * math.pi * u[:, 4]
    s = np.sqrt(np.maximum(0.0, 1.0 - z * z))
    d = np.stack([s * np.cos(phi), s * np.sin(phi), z], axis=1)
    return np.ascontiguousarray(o), np.ascontiguousarray(d), np.ascontiguousarray(u[:, 5])


def primary_rays(n_side: int, bounds):
    """Batch (A): coherent pinhole-camera rays looking at the scene box from outside."""
    lo = np.array([bounds[0], bounds[2], bounds[4]])
    hi = np.array([bounds[1], bounds[3], bounds[5]])
    ctr, ext = 0.5 * (lo + hi), float(np.max(hi - lo))
    eye = ctr + np.array([0.0, 0.0, 2.2 * ext])
    y, x = np.mgrid[0:n_side, 0:n_side]
    px = (x.reshape(-1) + 0.5) / n_side - 0.5
    py = 0.5 - (y.reshape(-1) + 0.5) / n_side
    d = np.stack([px * 1.1 * ext, py * 1.1 * ext, np.full(px.shape, -2.2 * ext)], axis=1)
    o = np.tile(eye, (len(px), 1))
    return np.ascontiguousarray(o), np.ascontiguousarray(d), np.zeros(len(px))
